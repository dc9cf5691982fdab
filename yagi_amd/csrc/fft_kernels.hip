// fft_kernels.hip -- batched complex FFT, Fft::run semantics (src/fft/mod.rs:39-57; the
// reference's arithmetic is the external crate rustfft 6.2, so this is a from-scratch engine
// for the same definition: unnormalised DFT, forward = e^{-j 2 pi n k / N}).
//
//   fft4096_kernel   config C3 (4096-pt x 65 536): one workgroup per transform, data makes one
//                    HBM round trip (64 KiB per transform = 16 B/point), everything else in
//                    registers + 34 KiB LDS (fft_core.hpp).
//   fft_n256m_kernel, fft8192_kernel   the same register scheme for 256..2048 and 8192
//   fft_pow2_kernel  other powers of two (N <= 128): LDS-staged register Stockham passes
//   fft_twopass_kernel  2^14 .. 2^16 points: column pass + row pass over the LDS Stockham core (two HBM round trips)
//   fft_mixed_kernel any other N <= 8192: mixed-radix Stockham passes in LDS over the plan's factor
//                    list, register butterflies for radix 16/8/4/2/3/5/7, direct R-term sums (exact
//                    table twiddles, index arithmetic mod N) for any other prime factor.
#include "fft_radix.hpp"
#include "kernels.hpp"

namespace yagi {

template <int SIGN>
__global__ void __launch_bounds__(256)
fft4096_kernel(const float2 *__restrict__ in, float2 *__restrict__ out,
               const float2 *__restrict__ tw, size_t batch) {
    __shared__ float2 lds[kFft4096LdsFloat2];
    {
        const size_t b = blockIdx.x;          // one transform per workgroup (grid = batch), no grid-stride loop
        (void)batch;
        const float2 *src = in + b * 4096;
        float2 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = ld_stream(src + 256 * a + threadIdx.x);
        fft4096_passes<SIGN>(v, lds, tw, out + b * 4096);
    }
}

// ---------------------------------------------------------------------------------------------
// N = 256 M, M in {1, 2, 4, 8} (256 .. 2048): the 4096-point scheme with a radix-M last pass (M = 1: none).
// A workgroup owns 16/M transforms (4096 points, 16 per lane); a transform is spread over 16 M lanes.
//   pass 1  lane b (< 16M)        : v[a]  = x[16M a + b]        -> Z_c[b] = W_N^{bc} sum_a v[a] W16^{ac}
//   pass 2  lane (c, b') = cM + b': v[a'] = Z_c[M a' + b']      -> U_{c,c'}[b'] = W_N^{16 b' c'} sum_a' ...
//   pass 3  lane u, pairs p = u + 16M i (p = c + 16c', i < 16/M): X[p + 256 d'] = sum_b' U_p[b'] W_M^{b' d'}
// Global loads and stores are runs of 16M consecutive points per transform.  LDS layouts (float2):
//   exchange 1: [transform][c][b], row stride 17M  (= M mod 32: pass-2 reads hit distinct banks)
//   exchange 2: [transform][b'][p], row stride 256 + 16/M (pass-2 writes hit distinct banks)
// M = 16 is fft4096_passes (fft_core.hpp).  Lanes of transforms past the end of the batch load zeros and
// skip their stores.
// ---------------------------------------------------------------------------------------------
template <int SIGN, int M>
__global__ void __launch_bounds__(256)
fft_n256m_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ tw,
                 size_t batch) {
    constexpr int N = 256 * M, LT = 16 * M, B = 16 / M;          // points, lanes per transform, transforms per WG
    __shared__ float2 lds[kFft4096LdsFloat2];
    const unsigned t = threadIdx.x, tr = t / LT, u = t % LT;
    const size_t g = (size_t)blockIdx.x * B + tr;
    const bool live = g < batch;
    const float2 *src = in + g * N;
    float2 v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a] = live ? ld_stream(src + LT * a + u) : make_float2(0.f, 0.f);
    fft_n256m_passes_to_regs<SIGN, M>(v, lds, tw);
    if (live) {
        float2 *dst = out + g * N + u;
#pragma unroll
        for (int i = 0; i < B; ++i)
#pragma unroll
            for (int d = 0; d < M; ++d) st_stream(dst + LT * i + 256 * d, v[i * M + d]);
    }
}

// ---------------------------------------------------------------------------------------------
// N = 8192 = 2 x 4096 (decimation in time): one 16-byte load per lane brings an even and an odd sample,
// the two 4096-point transforms run back to back in registers (fft4096_passes_to_regs), and the last
// radix-2 stage X[k] = E[k] + W_8192^k O[k], X[k + 4096] = E[k] - W_8192^k O[k] writes both halves
// coalesced.  `tw` = W_8192 table followed by the W_4096 table (capi.hip: make_twiddles half_too).
// ---------------------------------------------------------------------------------------------
template <int SIGN>
__global__ void __launch_bounds__(256)
fft8192_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ tw) {
    __shared__ float2 lds[kFft4096LdsFloat2];
    const unsigned t = threadIdx.x;
    const float4 *src = reinterpret_cast<const float4 *>(in + (size_t)blockIdx.x * 8192);
    const float2 *tw4096 = tw + 8192;
    float2 e[16], o[16];
#pragma unroll
    for (unsigned a = 0; a < 16; ++a) {
        const float4 q = src[256u * a + t];
        e[a] = make_float2(q.x, q.y);
        o[a] = make_float2(q.z, q.w);
    }
    fft4096_passes_to_regs<SIGN>(e, lds, tw4096);
    fft4096_passes_to_regs<SIGN>(o, lds, tw4096);
    float2 *dst = out + (size_t)blockIdx.x * 8192;
#pragma unroll
    for (unsigned d = 0; d < 16; ++d) {
        const unsigned k = t + 256u * d;
        const float2 p = cmul(o[d], tw[k]);
        dst[k] = cadd(e[d], p);
        dst[k + 4096] = csub(e[d], p);
    }
}

// Mixed radix, any N <= 8192 that is not a power of two: a workgroup owns nfr = 2048/N (>= 1) consecutive
// transforms; Stockham autosort passes over the plan's factor list (radices 16/8/4/2/3/5/7 as register
// butterflies, any other prime by direct sums), two LDS buffers.  One HBM round trip: 16 B/point.
template <int SIGN>
__global__ void __launch_bounds__(256)
fft_mixed_kernel(FftPlanDev p, const float2 *__restrict__ in, float2 *__restrict__ out, size_t batch, int nfr) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = p.n;
    float2 *bufA = reinterpret_cast<float2 *>(smem);
    float2 *bufB = bufA + (size_t)nfr * N;
    const float2 *tw = reinterpret_cast<const float2 *>(p.tw);
    const size_t b0 = (size_t)blockIdx.x * nfr;
    const int nb = (int)((batch - b0) < (size_t)nfr ? (batch - b0) : (size_t)nfr);
    const int total = nb * N;
    const float2 *gsrc = in + b0 * N;
    float2 *gdst = out + b0 * N;
    batched_for<256>(total, [&](int e) { return gsrc[e]; }, [&](int e, float2 v) { bufA[e] = v; });
    __syncthreads();
    float2 *src = bufA, *dst = bufB;
    int Ns = 1;
    for (int f = 0; f < p.nfac; ++f) {
        const int R = p.fac[f];
        switch (R) {
            case 16: stockham_pass_any<16, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 8: stockham_pass_any<8, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 7: stockham_pass_any<7, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 5: stockham_pass_any<5, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 4: stockham_pass_any<4, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 3: stockham_pass_any<3, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 2: stockham_pass_any<2, SIGN>(src, dst, N, Ns, nb, tw); break;
            default: stockham_pass_direct(src, dst, N, R, Ns, nb, tw); break;
        }
        __syncthreads();
        float2 *tmp = src; src = dst; dst = tmp;
        Ns *= R;
    }
    batched_for<256>(total, [&](int e) { return src[e]; }, [&](int e, float2 v) { gdst[e] = v; });
}

// Power-of-two N <= 8192 (other than 4096): a workgroup owns `nfr` consecutive transforms (so small
// N still fills 256 lanes), radix-16/8/4/2 register butterflies, Stockham autosort through two LDS
// buffers, twiddle table in LDS.  One HBM round trip: 16 B/point.
// FRFAST (2 <= nfr <= 32 transforms per workgroup): transform slots are `pitch` = N + 32/nfr apart and the
// passes map lanes transform-fastest, which makes every pass bank-conflict free (fft_radix.hpp); the
// global <-> LDS copies stay butterfly-fastest (coalesced).  A partial last group computes its missing
// transforms on whatever the LDS holds and does not store them.
template <int SIGN, bool FRFAST>
__global__ void __launch_bounds__(256)
fft_pow2_kernel(int N, Pow2Plan plan, const float2 *__restrict__ in, float2 *__restrict__ out,
                const float2 *__restrict__ tw, size_t batch, int nfr, int lgnfr, int pitch, int tw_in_lds) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *bufA = reinterpret_cast<float2 *>(smem);
    float2 *bufB = bufA + (size_t)nfr * pitch;
    // the plan's table already carries the direction's sign (tw_is_forward = false below);
    // it is copied to LDS unless the transform itself needs the room (N = 8192)
    const float2 *twl = tw;
    if (tw_in_lds) {
        float2 *t = bufB + (size_t)nfr * pitch;
        for (int e = threadIdx.x; e < N; e += 256) t[e] = tw[e];
        twl = t;
    }
    const int lgN = 31 - __builtin_clz((unsigned)N);
    const size_t b0 = (size_t)blockIdx.x * nfr;
    const int nb = (int)((batch - b0) < (size_t)nfr ? (batch - b0) : (size_t)nfr);
    const int total = nb * N;
    const float2 *src = in + b0 * N;
    float2 *dst = out + b0 * N;
    // global <-> LDS copies with the loads batched (devmath.hpp: batched_for)
    batched_for<256>(total, [&](int e) { return src[e]; },
                     [&](int e, float2 v) { bufA[(e >> lgN) * pitch + (e & (N - 1))] = v; });
    __syncthreads();
    const float2 *res = lds_fft_pow2<SIGN, FRFAST>(bufA, bufB, N, FRFAST ? nfr : nb, plan, twl, 1, false, pitch, lgnfr);
    batched_for<256>(total, [&](int e) { return res[(e >> lgN) * pitch + (e & (N - 1))]; },
                     [&](int e, float2 v) { dst[e] = v; });
}

// Bluestein element-wise stages: a[k] = x[k] w[k] zero-padded to m;  X[k] = w[k] y[k] / m
__global__ void __launch_bounds__(256)
bluestein_pre_kernel(const float2 *__restrict__ x, const float2 *__restrict__ w, int n, int m, size_t nb,
                     float2 *__restrict__ a) {
    const size_t total = nb * (size_t)m;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / m;
        const int k = (int)(e - b * m);
        a[e] = k < n ? cmul(x[b * n + k], w[k]) : make_float2(0.f, 0.f);
    }
}
__global__ void __launch_bounds__(256)
bluestein_post_kernel(const float2 *__restrict__ y, const float2 *__restrict__ w, int n, int m, size_t nb,
                      float2 *__restrict__ X) {
    const size_t total = nb * (size_t)n;
    const float inv = 1.0f / (float)m;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / n;
        const int k = (int)(e - b * n);
        X[e] = cscale(cmul(y[b * m + k], w[k]), inv);
    }
}
__global__ void __launch_bounds__(256)
bluestein_mul_kernel(float2 *__restrict__ f, const float2 *__restrict__ bf, int m, size_t nb) {
    const size_t total = nb * (size_t)m;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
        f[e] = cmul(f[e], bf[e & (size_t)(m - 1)]);
}

// ---------------------------------------------------------------------------------------------
// n = n1 n2 above 8192 points (n1, n2 <= 8192): four-step form.  x viewed as [n1][n2]:
//   A[n2][n1] = x^T;  B = FFT_n1 of each row;  C[k1][n2] = B[n2][k1] W_n^{n2 k1};  D = FFT_n2 of each row;
//   X[k2 n1 + k1] = D[k1][k2]  (third transpose).
// Five HBM round trips per point instead of one -- the price of not fitting a workgroup; the row transforms
// are the register kernels above.  The twiddle angle is evaluated in double per element (exact index n2 k1 < n).
// ---------------------------------------------------------------------------------------------
template <bool TWIDDLE>
__global__ void __launch_bounds__(256)
fft_transpose_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int rows, int cols,
                     const float2 *__restrict__ wlo, const float2 *__restrict__ whi) {
    __shared__ float2 tile[32][33];
    const size_t mat = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                // 32 x 8
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int r = r0 + ty + j, c = c0 + tx;
        if (r < rows && c < cols) {
            float2 v = in[mat + (size_t)r * cols + c];
            if (TWIDDLE) {                                  // W_n^{r c} = whi[m >> 12] wlo[m & 4095], m = r c < n
                const unsigned m = (unsigned)r * (unsigned)c;
                v = cmul(v, cmul(whi[m >> 12], wlo[m & 4095u]));
            }
            tile[ty + j][tx] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int c = c0 + ty + j, r = r0 + tx;
        if (r < rows && c < cols) out[mat + (size_t)c * rows + r] = tile[tx][ty + j];
    }
}

// ---------------------------------------------------------------------------------------------
// n = n1 n2, both powers of two <= 256 (2^14 .. 2^16 points): the four-step form in TWO launches, no transposes.
// x viewed as [n1][n2]:
//   pass 0 (columns): a workgroup takes nfr consecutive columns c0.. -- rows of nfr consecutive points in
//           memory -- transforms them over n1 and writes S[k1][c] = W_n^{k1 c} FFT_n1{x[.][c]}[k1] back in place;
//   pass 1 (rows):    a workgroup takes nfr consecutive rows k1 of S, transforms them over n2 and writes
//           X[k2 n1 + k1]: runs of nfr consecutive points again.
// Two HBM round trips instead of five.  Both passes run the LDS Stockham core transform-fastest (the nfr
// columns / rows are its transforms), so the strided side of each pass is what its lanes already do.
// ---------------------------------------------------------------------------------------------
template <int SIGN, int MODE>
__global__ void __launch_bounds__(256)
fft_twopass_kernel(int N, Pow2Plan plan, const float2 *__restrict__ in, float2 *__restrict__ out,
                   const float2 *__restrict__ tw, const float2 *__restrict__ wn, int n1, int n2, int nfr, int lgnfr,
                   int pitch) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *bufA = reinterpret_cast<float2 *>(smem);
    float2 *bufB = bufA + (size_t)nfr * pitch;
    float2 *twl = bufB + (size_t)nfr * pitch;
    for (int e = threadIdx.x; e < N; e += 256) twl[e] = tw[e];
    const size_t n = (size_t)n1 * n2;
    // pass 0: W_n^m = W_n^{256 (m >> 8)} W_n^{m & 255} from two 256-entry LDS tables (a per-point gather from the
    // n-entry table in global memory costs more than the transform: 64 distinct lines per wave access)
    float2 *wlo = twl + N, *whi = wlo + 256;
    if (MODE == 0) {
        wlo[threadIdx.x] = wn[threadIdx.x];
        whi[threadIdx.x] = wn[((size_t)threadIdx.x << 8) & (n - 1)];
    }
    const float2 *src = in + blockIdx.y * n;
    float2 *dst = out + blockIdx.y * n;
    const int t0 = blockIdx.x * nfr;                             // first column (pass 0) / row (pass 1) of the tile
    const int lgN = 31 - __builtin_clz((unsigned)N);
    // nfr * N = 2048 points: the eight loads of a lane are all issued before the first LDS write
    float2 r[8];
    if (MODE == 0) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = threadIdx.x + 256 * it, tr = e & (nfr - 1), p = e >> lgnfr;
            r[it] = src[(size_t)p * n2 + t0 + tr];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = threadIdx.x + 256 * it, tr = e & (nfr - 1), p = e >> lgnfr;
            bufA[tr * pitch + p] = r[it];
        }
    } else {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = threadIdx.x + 256 * it, tr = e >> lgN, p = e & (N - 1);
            r[it] = src[(size_t)(t0 + tr) * n2 + p];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = threadIdx.x + 256 * it, tr = e >> lgN, p = e & (N - 1);
            bufA[tr * pitch + p] = r[it];
        }
    }
    __syncthreads();
    const float2 *res = lds_fft_pow2<SIGN, true>(bufA, bufB, N, nfr, plan, twl, 1, false, pitch, lgnfr);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = threadIdx.x + 256 * it;
        const int tr = e & (nfr - 1), k = e >> lgnfr;
        float2 v = res[tr * pitch + k];
        if (MODE == 0) {
            const unsigned m = ((unsigned)k * (unsigned)(t0 + tr)) & (unsigned)(n - 1);
            dst[(size_t)k * n2 + t0 + tr] = cmul(v, cmul(whi[m >> 8], wlo[m & 255u]));
        } else {
            dst[(size_t)k * n1 + t0 + tr] = v;
        }
    }
}

static int launch_fft_two_pass(const FftPlanDev &p, const cf32 *in, cf32 *out, size_t batch, hipStream_t st) {
    const int n1 = p.fs_n1, n2 = p.fs_n2;
    const size_t n = (size_t)p.n;
    float2 *s0 = reinterpret_cast<float2 *>(p.fs_scratch);
    const float2 *wn = reinterpret_cast<const float2 *>(p.fs_wn);
    const bool fwd = p.dir == YAGI_FFT_FORWARD;
    auto pass = [&](int mode, int N, const float2 *tw, const float2 *src, float2 *dst, unsigned nb) -> int {
        const int nfr = 2048 / N;                                 // 16 columns / rows (128-byte runs) at N = 128, 8 at 256
        int lgnfr = 0;
        while ((1 << lgnfr) < nfr) ++lgnfr;
        const int pitch = frfast_pitch(N, nfr);
        const size_t lds = (2 * (size_t)nfr * pitch + (size_t)N + (mode == 0 ? 512 : 0)) * sizeof(float2);
        const Pow2Plan plan = make_pow2_plan(N);
        const dim3 grid((unsigned)((mode == 0 ? n2 : n1) / nfr), nb);
        if (mode == 0) {
            if (fwd) fft_twopass_kernel<-1, 0><<<grid, 256, lds, st>>>(N, plan, src, dst, tw, wn, n1, n2, nfr, lgnfr, pitch);
            else fft_twopass_kernel<+1, 0><<<grid, 256, lds, st>>>(N, plan, src, dst, tw, wn, n1, n2, nfr, lgnfr, pitch);
        } else {
            if (fwd) fft_twopass_kernel<-1, 1><<<grid, 256, lds, st>>>(N, plan, src, dst, tw, wn, n1, n2, nfr, lgnfr, pitch);
            else fft_twopass_kernel<+1, 1><<<grid, 256, lds, st>>>(N, plan, src, dst, tw, wn, n1, n2, nfr, lgnfr, pitch);
        }
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    };
    const size_t chunk = (size_t)p.fs_chunk * 2;                 // the whole scratch holds the one intermediate
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const unsigned nb = (unsigned)((batch - b0) < chunk ? (batch - b0) : chunk);
        const float2 *src = reinterpret_cast<const float2 *>(in) + b0 * n;
        float2 *dst = reinterpret_cast<float2 *>(out) + b0 * n;
        YG_TRY(pass(0, n1, reinterpret_cast<const float2 *>(p.fs_p1->tw), src, s0, nb));
        YG_TRY(pass(1, n2, reinterpret_cast<const float2 *>(p.fs_p2->tw), s0, dst, nb));
    }
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// Powers of two from 2^16 up: n = 256 n2 viewed as x[256][n2], the 256-point factor in registers
// (fft_n256m_passes_to_regs<SIGN, 1>), 16 transforms per workgroup, every global access a 128-byte run.
//   MODE 0 (columns): columns c0..c0+15 come in as 256 rows of 16 consecutive points, are turned through LDS into
//           the register layout, transformed, multiplied by W_n^{c k1} and written back the same way:
//           S[k1][c] = W_n^{k1 c} FFT_256{x[.][c]}[k1]
//   MODE 1 (rows, n2 = 256 only): rows k1_0..k1_0+15 of S are contiguous; their transforms leave transposed,
//           X[k2 * 256 + k1], again as runs of 16 points.
// For n2 > 256 the rows go through the n2-point plan and one tiled transposition (launch_fft_tile256).
// W_n^m comes from two exact tables (the lane needs five powers, the rest are products):
//   W_n^m = whi[m >> 12] * wlo[m & 4095],  wlo[j] = W_n^j, whi[j] = W_n^{4096 j}
// The column pass runs at what a read-HBM / write kernel reaches here (4.4 TB/s; prefetching the next tile's samples
// into registers, contiguous instead of strided rows and dropping the twiddles each changed nothing: r02_notes.md);
// the row pass reads what the column pass just wrote (Infinity Cache) and runs at 6.2 TB/s.
// Staging pitch 258 float2: the 16-column side of the exchange is conflict free, the transform side two-way.
// ---------------------------------------------------------------------------------------------
constexpr int kTilePitch = 258;
static_assert(16 * kTilePitch <= kFft4096LdsFloat2, "staging fits the exchange buffer");

__device__ __forceinline__ float2 wn_split(const float2 *__restrict__ wlo, const float2 *__restrict__ whi, unsigned m) {
    return cmul(whi[m >> 12], wlo[m & 4095u]);
}

template <int SIGN, int MODE>
__global__ void __launch_bounds__(256)
fft_tile256_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ tw256,
                   const float2 *__restrict__ wlo, const float2 *__restrict__ whi, int n2) {
    __shared__ float2 lds[kFft4096LdsFloat2];
    const unsigned t = threadIdx.x, tr = t >> 4, u = t & 15u;    // register layout: transform tr, lane u
    const unsigned col = t & 15u, r = t >> 4;                    // tile layout: 16 points of row r + 16 it
    const size_t mat = (size_t)blockIdx.y * 256 * (size_t)n2;
    const unsigned c0 = blockIdx.x * 16;
    float2 v[16];
    if (MODE == 0) {
        const float2 *src = in + mat + c0 + col + (size_t)r * n2;
        float2 s[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) s[it] = src[(size_t)(16 * it) * n2];
#pragma unroll
        for (int it = 0; it < 16; ++it) lds[col * kTilePitch + r + 16 * it] = s[it];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = lds[tr * kTilePitch + 16 * a + u];
        __syncthreads();
    } else {
        const float2 *src = in + mat + (size_t)(c0 + tr) * 256 + u;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = src[16 * a];
    }
    fft_n256m_passes_to_regs<SIGN, 1>(v, lds, tw256);             // v[i] = X[u + 16 i]
    if (MODE == 0) {
        const unsigned c = c0 + tr;                               // exponents c (u + 16 i) < n: no wrap
        float2 w[16];
        twiddle_powers_from(w, wn_split(wlo, whi, 16u * c), wn_split(wlo, whi, 32u * c), wn_split(wlo, whi, 64u * c),
                            wn_split(wlo, whi, 128u * c));
        const float2 e0 = wn_split(wlo, whi, c * u);
        v[0] = cmul(v[0], e0);
#pragma unroll
        for (int i = 1; i < 16; ++i) v[i] = cmul(v[i], cmul(w[i], e0));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) lds[tr * kTilePitch + u + 16 * i] = v[i];
    __syncthreads();
    const size_t pitch = MODE == 0 ? (size_t)n2 : 256;
    float2 *dst = out + mat + c0 + col + (size_t)r * pitch;
#pragma unroll
    for (int it = 0; it < 16; ++it) dst[(size_t)(16 * it) * pitch] = lds[col * kTilePitch + r + 16 * it];
}

template <int MODE>
static void launch_tile256(bool fwd, dim3 grid, const float2 *in, float2 *out, const float2 *tw256, const float2 *wlo,
                           const float2 *whi, int n2, hipStream_t st) {
    if (fwd) fft_tile256_kernel<-1, MODE><<<grid, 256, 0, st>>>(in, out, tw256, wlo, whi, n2);
    else fft_tile256_kernel<+1, MODE><<<grid, 256, 0, st>>>(in, out, tw256, wlo, whi, n2);
}

static int launch_fft_tile256(const FftPlanDev &p, const cf32 *in, cf32 *out, size_t batch, hipStream_t st) {
    const int n2 = p.fs_n2;
    const size_t n = (size_t)p.n;
    const FftPlanDev &f1 = *p.fs_p1, &f2 = *p.fs_p2;
    float2 *s0 = reinterpret_cast<float2 *>(p.fs_scratch), *s1 = s0 + (size_t)p.fs_chunk * n;
    const float2 *tw256 = reinterpret_cast<const float2 *>(f1.tw);
    const float2 *wlo = reinterpret_cast<const float2 *>(p.fs_wlo), *whi = reinterpret_cast<const float2 *>(p.fs_whi);
    const bool fwd = p.dir == YAGI_FFT_FORWARD;
    for (size_t b0 = 0; b0 < batch; b0 += (size_t)p.fs_chunk) {
        const unsigned nb = (unsigned)((batch - b0) < (size_t)p.fs_chunk ? (batch - b0) : (size_t)p.fs_chunk);
        const float2 *src = reinterpret_cast<const float2 *>(in) + b0 * n;
        float2 *dst = reinterpret_cast<float2 *>(out) + b0 * n;
        launch_tile256<0>(fwd, dim3((unsigned)(n2 / 16), nb), src, s0, tw256, wlo, whi, n2, st);
        YG_LAUNCH_CHECK();
        if (n2 == 256) {
            launch_tile256<1>(fwd, dim3(16, nb), s0, dst, tw256, wlo, whi, n2, st);
            YG_LAUNCH_CHECK();
        } else {
            YG_TRY(launch_fft_batch(f2, reinterpret_cast<const cf32 *>(s0), reinterpret_cast<cf32 *>(s1), (size_t)nb * 256, st));
            // D[k1][k2] -> X[k2][k1]
            fft_transpose_kernel<false><<<dim3((unsigned)(n2 / 32), 8, nb), 256, 0, st>>>(s1, dst, 256, n2, nullptr, nullptr);
            YG_LAUNCH_CHECK();
        }
    }
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// n = n1 n2 above 8192 points, any factors up to 1024 each (10 000 = 100 x 100, 48 000 = 240 x 200, 100 000 = 400 x 250):
// the four-step form in TWO launches, no transposes -- fft_twopass_kernel's scheme with the mixed-radix LDS passes of
// fft_mixed_kernel.  x viewed as [n1][n2]:
//   MODE 0 (columns): a workgroup takes nfr consecutive columns (rows of nfr consecutive points in memory: 128-byte
//           runs at nfr = 16, 64-byte runs at 8), transforms them over n1 and writes S[k1][c] = W_n^{k1 c} FFT_n1{x[.][c]}[k1]
//   MODE 1 (rows):    a workgroup takes nfr consecutive rows k1 of S (one contiguous piece), transforms them over n2 and
//           writes X[k2 n1 + k1]: runs of nfr points again.
// Two HBM round trips instead of the five of launch_fft_four_step.  W_n^m = whi[m >> 12] wlo[m & 4095] (exact split tables).
// ---------------------------------------------------------------------------------------------
template <int SIGN, int MODE>
__global__ void __launch_bounds__(256)
fft_mixed_twopass_kernel(FftPlanDev p, const float2 *__restrict__ in, float2 *__restrict__ out,
                         const float2 *__restrict__ wlo, const float2 *__restrict__ whi, int n1, int n2, int nfr) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = p.n;                                           // n1 (MODE 0) / n2 (MODE 1)
    float2 *bufA = reinterpret_cast<float2 *>(smem);
    float2 *bufB = bufA + (size_t)nfr * N;
    const float2 *tw = reinterpret_cast<const float2 *>(p.tw);
    const size_t n = (size_t)n1 * n2;
    const float2 *gsrc = in + blockIdx.y * n;
    float2 *gdst = out + blockIdx.y * n;
    const int t0 = blockIdx.x * nfr;                             // first column (MODE 0) / row (MODE 1) of the tile
    const int left = (MODE == 0 ? n2 : n1) - t0;
    const int nb = left < nfr ? left : nfr;
    const int total = nb * N;
    if (MODE == 0)
        batched_for<256>(total, [&](int e) { const int pp = e / nb, tr = e - pp * nb; return gsrc[(size_t)pp * n2 + t0 + tr]; },
                         [&](int e, float2 v) { const int pp = e / nb, tr = e - pp * nb; bufA[tr * N + pp] = v; });
    else
        batched_for<256>(total, [&](int e) { return gsrc[(size_t)t0 * n2 + e]; }, [&](int e, float2 v) { bufA[e] = v; });
    __syncthreads();
    float2 *src = bufA, *dst = bufB;
    int Ns = 1;
    for (int f = 0; f < p.nfac; ++f) {
        const int R = p.fac[f];
        switch (R) {
            case 16: stockham_pass_any<16, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 8: stockham_pass_any<8, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 7: stockham_pass_any<7, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 5: stockham_pass_any<5, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 4: stockham_pass_any<4, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 3: stockham_pass_any<3, SIGN>(src, dst, N, Ns, nb, tw); break;
            case 2: stockham_pass_any<2, SIGN>(src, dst, N, Ns, nb, tw); break;
            default: stockham_pass_direct(src, dst, N, R, Ns, nb, tw); break;
        }
        __syncthreads();
        float2 *tmp = src; src = dst; dst = tmp;
        Ns *= R;
    }
    const size_t opitch = MODE == 0 ? (size_t)n2 : (size_t)n1;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int k = e / nb, tr = e - k * nb;
        float2 v = src[tr * N + k];
        if (MODE == 0) {
            const unsigned m = (unsigned)k * (unsigned)(t0 + tr);       // < n1 n2
            v = cmul(v, cmul(whi[m >> 12], wlo[m & 4095u]));
        }
        gdst[(size_t)k * opitch + t0 + tr] = v;
    }
}

static int launch_fft_mixed_two_pass(const FftPlanDev &p, const cf32 *in, cf32 *out, size_t batch, hipStream_t st) {
    const int n1 = p.fs_n1, n2 = p.fs_n2;
    const size_t n = (size_t)p.n;
    float2 *s0 = reinterpret_cast<float2 *>(p.fs_scratch);
    const float2 *wlo = reinterpret_cast<const float2 *>(p.fs_wlo4), *whi = reinterpret_cast<const float2 *>(p.fs_whi4);
    const bool fwd = p.dir == YAGI_FFT_FORWARD;
    auto pass = [&](int mode, const FftPlanDev &f, const float2 *src, float2 *dst, unsigned nb) -> int {
        // columns / rows per workgroup: 16 up to N = 128, 8 up to 512 (16 there: 15-26 % slower at 48 000 / 100 000, LDS per
        // workgroup), 4 up to 1024 (32-byte runs, still ahead of the five-trip form: 360 000 -22 %, 600 000 -17 %, 10^6 -2 %)
        const int nfr = f.n <= 128 ? 16 : (f.n <= 512 ? 8 : 4);
        const size_t lds = 2 * (size_t)nfr * f.n * sizeof(float2);
        const dim3 grid((unsigned)(((mode == 0 ? n2 : n1) + nfr - 1) / nfr), nb);
        if (mode == 0) {
            if (fwd) fft_mixed_twopass_kernel<-1, 0><<<grid, 256, lds, st>>>(f, src, dst, wlo, whi, n1, n2, nfr);
            else fft_mixed_twopass_kernel<+1, 0><<<grid, 256, lds, st>>>(f, src, dst, wlo, whi, n1, n2, nfr);
        } else {
            if (fwd) fft_mixed_twopass_kernel<-1, 1><<<grid, 256, lds, st>>>(f, src, dst, wlo, whi, n1, n2, nfr);
            else fft_mixed_twopass_kernel<+1, 1><<<grid, 256, lds, st>>>(f, src, dst, wlo, whi, n1, n2, nfr);
        }
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    };
    const size_t chunk = (size_t)p.fs_chunk * 2;                 // the whole scratch holds the one intermediate (half of it: 3-8 % slower)
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const unsigned nb = (unsigned)((batch - b0) < chunk ? (batch - b0) : chunk);
        const float2 *src = reinterpret_cast<const float2 *>(in) + b0 * n;
        float2 *dst = reinterpret_cast<float2 *>(out) + b0 * n;
        YG_TRY(pass(0, *p.fs_p1, src, s0, nb));
        YG_TRY(pass(1, *p.fs_p2, s0, dst, nb));
    }
    return YAGI_OK;
}

static int launch_fft_four_step(const FftPlanDev &p, const cf32 *in, cf32 *out, size_t batch, hipStream_t st) {
    const int n1 = p.fs_n1, n2 = p.fs_n2;
    const size_t n = (size_t)p.n;
    const FftPlanDev &f1 = *p.fs_p1, &f2 = *p.fs_p2;
    float2 *s0 = reinterpret_cast<float2 *>(p.fs_scratch), *s1 = s0 + (size_t)p.fs_chunk * n;
    const float2 *wlo = reinterpret_cast<const float2 *>(p.fs_wlo4), *whi = reinterpret_cast<const float2 *>(p.fs_whi4);
    const unsigned g1 = (unsigned)((n1 + 31) / 32), g2 = (unsigned)((n2 + 31) / 32);
    for (size_t b0 = 0; b0 < batch; b0 += (size_t)p.fs_chunk) {
        const unsigned nb = (unsigned)((batch - b0) < (size_t)p.fs_chunk ? (batch - b0) : (size_t)p.fs_chunk);
        const float2 *src = reinterpret_cast<const float2 *>(in) + b0 * n;
        float2 *dst = reinterpret_cast<float2 *>(out) + b0 * n;
        // x[n1][n2] -> A[n2][n1]
        fft_transpose_kernel<false><<<dim3(g2, g1, nb), 256, 0, st>>>(src, s0, n1, n2, nullptr, nullptr);
        YG_LAUNCH_CHECK();
        YG_TRY(launch_fft_batch(f1, reinterpret_cast<const cf32 *>(s0), reinterpret_cast<cf32 *>(s1), (size_t)nb * n2, st));
        // B[n2][k1] -> C[k1][n2] with W_n^{n2 k1}
        fft_transpose_kernel<true><<<dim3(g1, g2, nb), 256, 0, st>>>(s1, s0, n2, n1, wlo, whi);
        YG_LAUNCH_CHECK();
        YG_TRY(launch_fft_batch(f2, reinterpret_cast<const cf32 *>(s0), reinterpret_cast<cf32 *>(s1), (size_t)nb * n1, st));
        // D[k1][k2] -> X[k2][k1]
        fft_transpose_kernel<false><<<dim3(g2, g1, nb), 256, 0, st>>>(s1, dst, n1, n2, nullptr, nullptr);
        YG_LAUNCH_CHECK();
    }
    return YAGI_OK;
}

// Bluestein with m = 4096 or 8192 (1024 < n <= 4096 with a prime factor above 89) in ONE kernel: a workgroup owns one
// transform -- x w zero-padded to m, FFT_m, times FFT_m{b}, inverse FFT_m, times w / m -- with the m-point transforms
// chained in registers (fft4096_passes_to_regs leaves X[t + 256 d] where its first pass expects x[256 a + t]).
// n points in, n points out: 16 B per point instead of the ~240 of the five launches; the chirp, FFT{b} and twiddle
// tables come from L2.  m = 8192: forward by decimation in time (even / odd samples -> two 4096-point transforms ->
// X[k], X[k + 4096]), inverse by decimation in frequency (Y[k] +- Y[k + 4096] -> two 4096-point transforms -> even / odd
// samples), so the register layouts meet without an exchange.
template <int M8K>
__global__ void __launch_bounds__(256)
bluestein_fused_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, const float2 *__restrict__ w,
                       const float2 *__restrict__ bf, const float2 *__restrict__ twf, const float2 *__restrict__ twb, int n) {
    __shared__ float2 lds[kFft4096LdsFloat2];
    const unsigned t = threadIdx.x;
    const float2 *src = in + (size_t)blockIdx.x * n;
    float2 *dst = out + (size_t)blockIdx.x * n;
    const float2 zero = make_float2(0.f, 0.f);
    if constexpr (!M8K) {
        const float inv = 1.0f / 4096.0f;
        float2 v[16];
#pragma unroll
        for (unsigned a = 0; a < 16; ++a) {
            const unsigned k = 256u * a + t;
            v[a] = (int)k < n ? cmul(src[k], w[k]) : zero;
        }
        fft4096_passes_to_regs<-1>(v, lds, twf);
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) v[d] = cmul(v[d], bf[t + 256u * d]);
        fft4096_passes_to_regs<+1>(v, lds, twb);
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) {
            const unsigned k = t + 256u * d;
            if ((int)k < n) dst[k] = cscale(cmul(v[d], w[k]), inv);
        }
    } else {
        const float inv = 1.0f / 8192.0f;
        const float2 *twf4 = twf + 8192, *twb4 = twb + 8192;     // the W_4096 tables behind the W_8192 ones
        float2 e[16], o[16];
#pragma unroll
        for (unsigned a = 0; a < 16; ++a) {
            const unsigned k = 2u * (256u * a + t);
            e[a] = (int)k < n ? cmul(src[k], w[k]) : zero;
            o[a] = (int)(k + 1) < n ? cmul(src[k + 1], w[k + 1]) : zero;
        }
        fft4096_passes_to_regs<-1>(e, lds, twf4);
        fft4096_passes_to_regs<-1>(o, lds, twf4);
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) {
            const unsigned k = t + 256u * d;
            const float2 p = cmul(o[d], twf[k]);
            const float2 y0 = cmul(cadd(e[d], p), bf[k]), y1 = cmul(csub(e[d], p), bf[k + 4096]);
            e[d] = cadd(y0, y1);
            o[d] = cmul(csub(y0, y1), twb[k]);
        }
        fft4096_passes_to_regs<+1>(e, lds, twb4);
        fft4096_passes_to_regs<+1>(o, lds, twb4);
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) {
            const unsigned k = 2u * (t + 256u * d);
            if ((int)k < n) dst[k] = cscale(cmul(e[d], w[k]), inv);
            if ((int)(k + 1) < n) dst[k + 1] = cscale(cmul(o[d], w[k + 1]), inv);
        }
    }
}

static int launch_fft_bluestein(const FftPlanDev &p, const cf32 *in, cf32 *out, size_t batch, hipStream_t st) {
    const int m = p.bs_m;
    if ((m == 4096 || m == 8192) && !p.bs_fwd) {
        if (batch > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "batch too large");
        const float2 *fin = reinterpret_cast<const float2 *>(in), *w = reinterpret_cast<const float2 *>(p.bs_w);
        const float2 *bf = reinterpret_cast<const float2 *>(p.bs_bf);
        const float2 *twf = reinterpret_cast<const float2 *>(p.bs_twf), *twb = reinterpret_cast<const float2 *>(p.bs_twb);
        float2 *fout = reinterpret_cast<float2 *>(out);
        if (m == 4096) bluestein_fused_kernel<0><<<(unsigned)batch, 256, 0, st>>>(fin, fout, w, bf, twf, twb, p.n);
        else bluestein_fused_kernel<1><<<(unsigned)batch, 256, 0, st>>>(fin, fout, w, bf, twf, twb, p.n);
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    }
    FftPlanDev fwd, bwd;
    if (p.bs_fwd && p.bs_bwd) {          // m > 8192: full plans with their own (four-step) resources
        fwd = *p.bs_fwd;
        bwd = *p.bs_bwd;
    } else {
        fwd.n = bwd.n = m;
        fwd.dir = YAGI_FFT_FORWARD;
        bwd.dir = YAGI_FFT_BACKWARD;
        fwd.tw = p.bs_twf;
        bwd.tw = p.bs_twb;
    }
    float2 *s0 = reinterpret_cast<float2 *>(p.bs_scratch), *s1 = s0 + (size_t)p.bs_chunk * m;
    const float2 *w = reinterpret_cast<const float2 *>(p.bs_w);
    for (size_t b0 = 0; b0 < batch; b0 += (size_t)p.bs_chunk) {
        const size_t nb = (batch - b0) < (size_t)p.bs_chunk ? (batch - b0) : (size_t)p.bs_chunk;
        size_t g = (nb * (size_t)m + 255) / 256;
        if (g > 16384) g = 16384;
        bluestein_pre_kernel<<<(unsigned)g, 256, 0, st>>>(reinterpret_cast<const float2 *>(in) + b0 * p.n, w, p.n, m, nb, s0);
        YG_LAUNCH_CHECK();
        YG_TRY(launch_fft_batch(fwd, reinterpret_cast<const cf32 *>(s0), reinterpret_cast<cf32 *>(s1), nb, st));
        bluestein_mul_kernel<<<(unsigned)g, 256, 0, st>>>(s1, reinterpret_cast<const float2 *>(p.bs_bf), m, nb);
        YG_LAUNCH_CHECK();
        YG_TRY(launch_fft_batch(bwd, reinterpret_cast<const cf32 *>(s1), reinterpret_cast<cf32 *>(s0), nb, st));
        bluestein_post_kernel<<<(unsigned)g, 256, 0, st>>>(s0, w, p.n, m, nb, reinterpret_cast<float2 *>(out) + b0 * p.n);
        YG_LAUNCH_CHECK();
    }
    return YAGI_OK;
}

int launch_fft_batch(const FftPlanDev &p, const cf32 *in, cf32 *out, size_t batch, hipStream_t st) {
    if (batch == 0) return YAGI_OK;
    const float2 *fin = reinterpret_cast<const float2 *>(in);
    float2 *fout = reinterpret_cast<float2 *>(out);
    const float2 *tw = reinterpret_cast<const float2 *>(p.tw);
    if (p.n == 4096) {
        if (batch > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "batch too large");
        const unsigned grid = (unsigned)batch;
        if (p.dir == YAGI_FFT_FORWARD)
            fft4096_kernel<-1><<<grid, 256, 0, st>>>(fin, fout, tw, batch);
        else
            fft4096_kernel<+1><<<grid, 256, 0, st>>>(fin, fout, tw, batch);
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    }
    if (p.bs_m) return launch_fft_bluestein(p, in, out, batch, st);
    if (p.fs_n1 && p.fs_wlo) return launch_fft_tile256(p, in, out, batch, st);
    if (p.fs_n1 && p.fs_wn) return launch_fft_two_pass(p, in, out, batch, st);
    if (p.fs_n1 && p.fs_n1 <= kFftTwoPassMixedMax && p.fs_n2 <= kFftTwoPassMixedMax && p.fs_p1->nfac && p.fs_p2->nfac)
        return launch_fft_mixed_two_pass(p, in, out, batch, st);
    if (p.fs_n1) return launch_fft_four_step(p, in, out, batch, st);
    if (p.n > kFftMaxLds) return fail(YAGI_ERR_INTERNAL, "fft size %d has no plan resources", p.n);
    if (p.n == 8192) {
        if (batch > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "batch too large");
        if (p.dir == YAGI_FFT_FORWARD) fft8192_kernel<-1><<<(unsigned)batch, 256, 0, st>>>(fin, fout, tw);
        else fft8192_kernel<+1><<<(unsigned)batch, 256, 0, st>>>(fin, fout, tw);
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    }
    if (p.n == 256 || p.n == 512 || p.n == 1024 || p.n == 2048) {
        constexpr int kPts = 4096;
        const size_t per_wg = (size_t)(kPts / p.n);
        const size_t groups = (batch + per_wg - 1) / per_wg;
        if (groups > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "batch too large");
        const unsigned grid = (unsigned)groups;
        const bool fwd = p.dir == YAGI_FFT_FORWARD;
#define YG_N256M(MM)                                                                                 \
    do {                                                                                             \
        if (fwd) fft_n256m_kernel<-1, MM><<<grid, 256, 0, st>>>(fin, fout, tw, batch);               \
        else fft_n256m_kernel<+1, MM><<<grid, 256, 0, st>>>(fin, fout, tw, batch);                   \
    } while (0)
        if (p.n == 256) YG_N256M(1);
        else if (p.n == 512) YG_N256M(2);
        else if (p.n == 1024) YG_N256M(4);
        else YG_N256M(8);
#undef YG_N256M
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    }
    if (p.n >= 2 && (p.n & (p.n - 1)) == 0) {
        // 2048 points per workgroup (<= 32 transforms) for N <= 2048: 2 x 16 KiB of LDS + table, 3-4 workgroups per CU
        int nfr = (p.n <= 2048 ? 2048 : 4096) / p.n;
        if (nfr < 1) nfr = 1;
        if (nfr > 32) nfr = 32;
        const bool frfast = nfr >= 2;
        int lgnfr = 0;
        while ((1 << lgnfr) < nfr) ++lgnfr;
        const int pitch = frfast ? frfast_pitch(p.n, nfr) : p.n;
        const int tw_in_lds = p.n <= 4096 ? 1 : 0;
        const size_t lds2 = (2 * (size_t)nfr * pitch + (tw_in_lds ? (size_t)p.n : 0)) * sizeof(float2);
        const size_t groups = (batch + nfr - 1) / nfr;
        if (groups > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "batch too large");
        const unsigned grid2 = (unsigned)groups;
        const bool fwd = p.dir == YAGI_FFT_FORWARD;
        const void *fn = frfast ? (fwd ? reinterpret_cast<const void *>(fft_pow2_kernel<-1, true>)
                                       : reinterpret_cast<const void *>(fft_pow2_kernel<+1, true>))
                                : (fwd ? reinterpret_cast<const void *>(fft_pow2_kernel<-1, false>)
                                       : reinterpret_cast<const void *>(fft_pow2_kernel<+1, false>));
        if (lds2 > 64 * 1024) YG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const Pow2Plan plan = make_pow2_plan(p.n);
#define YG_POW2_LAUNCH(S, F) fft_pow2_kernel<S, F><<<grid2, 256, lds2, st>>>(p.n, plan, fin, fout, tw, batch, nfr, lgnfr, pitch, tw_in_lds)
        if (frfast) { if (fwd) YG_POW2_LAUNCH(-1, true); else YG_POW2_LAUNCH(+1, true); }
        else { if (fwd) YG_POW2_LAUNCH(-1, false); else YG_POW2_LAUNCH(+1, false); }
#undef YG_POW2_LAUNCH
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    }
    int nfr = 2048 / p.n;
    if (nfr < 1) nfr = 1;
    const size_t lds = 2 * (size_t)nfr * p.n * sizeof(float2);
    const size_t groups = (batch + nfr - 1) / nfr;
    if (groups > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "batch too large");
    const bool fwd = p.dir == YAGI_FFT_FORWARD;
    if (lds > 64 * 1024) {
        const void *fn = fwd ? reinterpret_cast<const void *>(fft_mixed_kernel<-1>)
                             : reinterpret_cast<const void *>(fft_mixed_kernel<+1>);
        YG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (fwd) fft_mixed_kernel<-1><<<(unsigned)groups, 256, lds, st>>>(p, fin, fout, batch, nfr);
    else fft_mixed_kernel<+1><<<(unsigned)groups, 256, lds, st>>>(p, fin, fout, batch, nfr);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// Fft::shift (fft/mod.rs:50-57): swap the two halves; for odd n the last element stays.
__global__ void __launch_bounds__(256) fft_shift_kernel(float2 *buf, size_t n, size_t batch) {
    const size_t n2 = n / 2;       // (n-1)/2 for odd n == n/2 in integer arithmetic
    const size_t total = n2 * batch;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / n2, i = e - b * n2;
        float2 *v = buf + b * n;
        const float2 t = v[i];
        v[i] = v[i + n2];
        v[i + n2] = t;
    }
}

int launch_fft_shift(cf32 *buf, size_t n, size_t batch, hipStream_t st) {
    const size_t total = (n / 2) * batch;
    if (total == 0) return YAGI_OK;
    size_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    fft_shift_kernel<<<(unsigned)g, 256, 0, st>>>(reinterpret_cast<float2 *>(buf), n, batch);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

}  // namespace yagi
