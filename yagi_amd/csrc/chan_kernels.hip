// chan_kernels.hip -- polyphase channelizers multichannel::firpfbch / firpfbch2: analyzers and synthesizers (tiled, column-sliding and wide forms).
//
// The reference module is EMPTY (src/multichannel/mod.rs, 0 lines; LIQUID_COMPAT.md:1765-1798),
// so these kernels implement liquid-dsp's published semantics composed from the reference's own
// pieces: the FirPfb branch split h_i[n] = h[i + n*M] (firpfb.rs:45-52), Window state
// (window.rs:77-85), dotprod (dotprod/mod.rs:33-45) and the unnormalised DFT of Fft
// (fft/mod.rs:19-26).  See oracle/yagi_oracle.c for the sequential restatement they are tested
// against (parity unpinned by the reference; property tests in tests/test_chan_*.py).
//
// Closed forms used here (X = hist ++ x, X[0] = x[0], f = frame / s = step index of this call):
//   firpfbch  : V[f][c] = sum_{n<p} h[(M-1-c) + n*M] * X[(f-n)*M + c]          c in [0,M)
//               y[f][k] = sum_c V[f][c] e^{-j 2 pi c k / M}
//   firpfbch2 : window b is fed on steps of parity (b >= M/2), at position
//               pos_b = (b < M/2 ? M/2-1-b : M-1-b); with s_b = latest such step <= s and
//               i = (b - (s odd ? M/2 : 0)) mod M:
//               V[s][b] = sum_{n<2m} h[i + n*M] * X[s_b*M/2 + pos_b - n*M]
//               y[s][k] = (1/M) sum_b V[s][b] e^{+j 2 pi b k / M}
//   sharded   : rank r of R keeps k = r + R*q:  fold Z[b'] = e^{+j2pi b' r/M} sum_a e^{+j2pi a r/R}
//               V[(M/R)a + b'], then an (M/R)-point inverse DFT over b' (decimation in frequency),
//               so each GPU writes only M/R outputs per step.
// One workgroup owns a tile of frames.  firpfbch (small M, p up to 16): input tile + history staged in
// LDS once (8 B/sample from HBM), branch dot products from LDS; firpfbch2 (M = 256: the history is
// ~4 tiles long) reads samples straight from L1/L2 instead (measured faster, profiles/r01_notes.md); the M-point DFTs as Stockham passes in LDS, coalesced
// [frame][channel] stores (8 or 16 B per input sample).  HBM-bound by construction.
#include <type_traits>

#include "fft_radix.hpp"
#include "kernels.hpp"

namespace yagi {

// Cache policy of the analyzers' streamed accesses (interleaved A/B, profiles/r03_notes.md): sample loads sc1 (L1
// bypassed) -1.4 ... -1.6 %; channel stores stay plain -- their 64- to 128-byte runs per wave instruction lose 2.8 ... 3.6 %
// as non-temporal stores (the synthesizers, whose stores are whole lines, gain 2 % and use the library default).
constexpr int kAnaLoad = kStreamLoad, kAnaStore = 0;

struct FacList { int n; int f[16]; };

static FacList factorize_small(int n) {
    FacList L{0, {0}};
    while (n % 4 == 0) { L.f[L.n++] = 4; n /= 4; }
    for (int p : {2, 3, 5, 7}) while (n % p == 0) { L.f[L.n++] = p; n /= p; }
    for (int p = 11; n > 1 && L.n < 15; p += 2) while (n % p == 0 && L.n < 15) { L.f[L.n++] = p; n /= p; }
    if (n > 1) L.f[L.n++] = n;
    return L;
}

// nfr independent N-point DFTs stored [frame][N] in LDS buffer `src`; result buffer returned.
// tw = W_Mtab^m table (forward sign), N * tw_scale == Mtab; conj => inverse transform.
// Radices 2/3/4/5/7 run as register butterflies (fft_radix.hpp), any other prime factor as direct sums.
template <bool CONJ>
__device__ __forceinline__ float2 *lds_dft_frames_t(float2 *src, float2 *dst, int N, int nfr,
                                                    const FacList &fl, const float2 *__restrict__ tw,
                                                    int tw_scale) {
    constexpr int SIGN = CONJ ? +1 : -1;
    int Ns = 1;
    const int total = N * nfr;
    for (int f = 0; f < fl.n; ++f) {
        const int R = fl.f[f];
        const int T = N / R;
        const int tw_k = N / (Ns * R);
        switch (R) {
            case 2: stockham_pass_any<2, SIGN>(src, dst, N, Ns, nfr, tw, tw_scale, CONJ); break;
            case 3: stockham_pass_any<3, SIGN>(src, dst, N, Ns, nfr, tw, tw_scale, CONJ); break;
            case 4: stockham_pass_any<4, SIGN>(src, dst, N, Ns, nfr, tw, tw_scale, CONJ); break;
            case 5: stockham_pass_any<5, SIGN>(src, dst, N, Ns, nfr, tw, tw_scale, CONJ); break;
            case 7: stockham_pass_any<7, SIGN>(src, dst, N, Ns, nfr, tw, tw_scale, CONJ); break;
            default:
                for (int e = threadIdx.x; e < total; e += blockDim.x) {
                    const int fr = e / N, idx = e - fr * N;
                    const int q = idx / T, j = idx - q * T;
                    const int k = j % Ns;
                    const int step = (k * tw_k + q * T) % N;
                    const float2 *s = src + fr * N;
                    float2 acc = s[j];
                    int m = 0;
                    for (int r = 1; r < R; ++r) {
                        m += step;
                        if (m >= N) m -= N;
                        float2 w = tw[m * tw_scale];
                        if (CONJ) w.y = -w.y;
                        acc = cadd(acc, cmul(s[j + r * T], w));
                    }
                    dst[fr * N + (j / Ns) * Ns * R + k + q * Ns] = acc;
                }
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
        Ns *= R;
    }
    return src;
}
__device__ __forceinline__ float2 *lds_dft_frames(float2 *src, float2 *dst, int N, int nfr,
                                                  const FacList &fl, const float2 *__restrict__ tw,
                                                  int tw_scale, bool conj) {
    return conj ? lds_dft_frames_t<true>(src, dst, N, nfr, fl, tw, tw_scale)
                : lds_dft_frames_t<false>(src, dst, N, nfr, fl, tw, tw_scale);
}

__device__ __forceinline__ float2 load_hist(const float2 *__restrict__ hist, int hist_len,
                                            const float2 *__restrict__ x, long long idx, long long x_len) {
    if (idx >= 0) return (idx < x_len) ? x[idx] : make_float2(0.f, 0.f);
    const long long h = hist_len + idx;
    return (h >= 0) ? hist[h] : make_float2(0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------
// firpfbch analyzer
// ---------------------------------------------------------------------------------------------
// POW2: M is a power of two -> register-butterfly Stockham passes (fft_radix.hpp); otherwise the
// general one-output-per-lane passes.  Taps (transposed, [n][c]) and the W_M table live in LDS.
template <bool POW2>
__global__ void __launch_bounds__(256)
firpfbch_kernel(const float2 *__restrict__ hist, const float2 *__restrict__ x,
                const float *__restrict__ h, int M, int p, const float2 *__restrict__ twM,
                FacList fl, Pow2Plan plan, float2 *__restrict__ y, size_t nframes, int F) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);              // (F + p - 1) * M samples
    float2 *va = xs + (size_t)(F + p - 1) * M;                  // F * M
    float2 *vb = va + (size_t)F * M;                            // F * M
    float2 *twl = vb + (size_t)F * M;                           // M
    float *hs = reinterpret_cast<float *>(twl + M);             // p * M : hs[n*M + c] = h[(M-1-c) + n*M]
    const int hist_len = (p - 1) * M;
    for (int e = threadIdx.x; e < M; e += 256) twl[e] = twM[e];
    for (int e = threadIdx.x; e < p * M; e += 256) {
        const int n = e / M, c = e - n * M;
        hs[e] = h[(M - 1 - c) + n * M];
    }
    for (size_t tile = blockIdx.x; tile * F < nframes; tile += gridDim.x) {
        const size_t f0 = tile * F;
        const int nf = (int)((nframes - f0) < (size_t)F ? (nframes - f0) : (size_t)F);
        const long long base = (long long)f0 * M - hist_len;
        const int nspan = (nf + p - 1) * M;
        const long long x_len = (long long)nframes * M;
        if (base >= 0 && base + nspan <= x_len) {               // interior tile: straight coalesced copy
            const float2 *src = x + base;
            for (int u = threadIdx.x; u < nspan; u += 256) xs[u] = src[u];
        } else {
            for (int u = threadIdx.x; u < nspan; u += 256) xs[u] = load_hist(hist, hist_len, x, base + u, x_len);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nf * M; e += 256) {
            const int f = e / M, c = e - f * M;
            float2 acc = make_float2(0.f, 0.f);
            // X[(f0+f-n)*M + c] sits at span offset (f + p-1 - n)*M + c
            const float2 *xp = xs + (f + p - 1) * M + c;
            for (int n = 0; n < p; ++n) {
                const float hv = hs[n * M + c];
                const float2 sv = xp[-n * M];
                acc.x = fmaf(sv.x, hv, acc.x);
                acc.y = fmaf(sv.y, hv, acc.y);
            }
            va[e] = acc;
        }
        __syncthreads();
        float2 *res = POW2 ? lds_fft_pow2<-1>(va, vb, M, nf, plan, twl, 1)
                           : lds_dft_frames(va, vb, M, nf, fl, twl, 1, false);
        for (int e = threadIdx.x; e < nf * M; e += 256) y[f0 * M + e] = res[e];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// firpfbch, column-sliding form (M in {64,128,256}, p in {4,8,16}): every lane owns ONE branch column c
// and walks a run of consecutive frames with its p taps and the last p samples of that column in
// registers (one coalesced 8-byte load + p packed FMAs per frame and lane, no LDS in the FIR part).
// A workgroup = 256/M column groups, each on its own run of frames; per tile of 16 frames the branch
// outputs go to LDS once, the workgroup runs the M-point register-butterfly passes over all 256/M*16
// transforms together and stores [frame][channel] fully coalesced.
// HBM traffic: 8 B/sample in (+ (p-1)/run halo, L2-served between the groups of one workgroup) + 8 out.
// History after the block = last hist_len samples of (hist ++ x[0..n_in)), written by the LAST workgroup of the
// analyzer kernel itself into the object's other history buffer (saves the separate 4-5 us update launch).
__device__ __forceinline__ void chan_write_next_hist(const float2 *__restrict__ hist, int hist_len,
                                                     const float2 *__restrict__ x, size_t n_in,
                                                     float2 *__restrict__ hist_next) {
    if (hist_next == nullptr || blockIdx.x != gridDim.x - 1) return;
    for (int j = threadIdx.x; j < hist_len; j += blockDim.x) {
        const size_t c = n_in + (size_t)j;
        hist_next[j] = (c < (size_t)hist_len) ? hist[c] : x[c - (size_t)hist_len];
    }
}

// ---------------------------------------------------------------------------------------------
// LDS slot pitch of one transform in the column kernels: M + 32/nq for nq <= 32 transforms in flight
// (fft_radix.hpp), M + 1 (odd: the lanes of a pass, one transform apart, cover all banks) beyond
__host__ __device__ constexpr int col_pitch(int M, int nq) { return M + (nq <= 32 ? 32 / nq : 1); }
constexpr int kColTile = 16;
constexpr int kColHalf = 8;
constexpr size_t kColWgs = 1024;               // workgroups a launch aims for: four per CU

// M = 2^LGM is a template parameter so that only the two radix passes M needs are instantiated (with a run-time
// plan the register allocation is that of the widest radix: 200 VGPRs).
// FULL: every workgroup of the launch holds G * run frames (otherwise the whole launch takes FULL = false).
template <int P, int LGM, bool FULL>
__global__ void __launch_bounds__(256)
firpfbch_col_kernel(const float2 *__restrict__ hist, const float2 *__restrict__ x,
                    const float *__restrict__ h, const float2 *__restrict__ twM,
                    float2 *__restrict__ y, size_t nframes, int run, float2 *__restrict__ hist_next,
                    int p_real /* taps per branch, <= P: the rest are zero */) {
    constexpr int M = 1 << LGM, lgM = LGM;
    chan_write_next_hist(hist, (p_real - 1) * M, x, nframes * (size_t)M, hist_next);
    // 8, 16: one pass; 32 = 8 x 4, 64 = 8 x 8, 128 = 16 x 8, 256 = 16 x 16
    constexpr int R0 = (LGM == 3 || LGM == 5 || LGM == 6) ? 8 : 16, R1 = M / R0;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int G = 256 / M;
    constexpr int nq = G * kColHalf, lgnq = 8 - LGM + 3;        // transforms per half tile (kColHalf = 8)
    constexpr int pitch = col_pitch(M, nq);
    float2 *va = reinterpret_cast<float2 *>(smem);              // [256/M groups][8 frames][pitch]
    float2 *vb = va + nq * pitch;
    float2 *twl = vb + nq * pitch;                              // M
    const int g = threadIdx.x >> lgM, c = threadIdx.x & (M - 1);
    for (int e = threadIdx.x; e < M; e += 256) twl[e] = twM[e];
    float hc[P];
#pragma unroll
    for (int n = 0; n < P; ++n) hc[n] = n < p_real ? h[(M - 1 - c) + n * M] : 0.0f;   // branch lengths between the built sizes: zero-padded
    const int hist_len = (p_real - 1) * M;
    const long long x_len = (long long)nframes * M;
    // one workgroup = G consecutive runs of `run` frames (no grid-stride loop); all frame indices below are
    // 32-bit offsets from the workgroup's first frame
    const long long wg_first = (long long)blockIdx.x * G * run;
    const long long wg_left = (long long)nframes - wg_first;                     // > 0
    const int wg_frames = (int)(wg_left < (long long)G * run ? wg_left : (long long)G * run);
    auto group_frames = [&](int gq) {                // frames of group gq that exist
        const int v = wg_frames - gq * run;
        return v < 0 ? 0 : (v > run ? run : v);
    };
    const long long f_begin = wg_first + (long long)g * run;
    const int nvalid = group_frames(g);
    const float2 *xg = x + f_begin * M + c;          // dereferenced for existing frames only
    float2 w[P];                                     // ring: frame f at slot (f - f_begin) mod P
#pragma unroll
    for (int n = 1; n < P; ++n)
        w[P - n] = load_hist(hist, hist_len, x, (f_begin - n) * M + c, x_len);
    float2 xa[kColHalf], xb[kColHalf];               // the next two half tiles of this column, in flight
    // M = 64 with 16 taps: ONE half tile in flight.  With two, the kernel holds 150 VGPRs = three waves per SIMD; with one,
    // 124 = four, and at 4096 workgroups (four rounds of the 1024 resident ones) it measures 214.9 us against 227.9 us
    // for two in flight at 2048 workgroups (1024 / 2048 workgroups with one in flight: 233.0 / 235.6 us)
    constexpr bool kOneInFlight = LGM == 6 && P == 16;
    constexpr int kAhead = kOneInFlight ? kColHalf : kColTile;
    // A full workgroup addresses its samples and outputs through buffer descriptors: one VGPR byte offset per lane,
    // the frame steps in SGPRs -- no 64-bit address arithmetic, no per-frame checks.
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + wg_first * M, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + wg_first * M, 0xffffffffu);
    const unsigned vx = 8u * ((unsigned)(g * run) * M + c);
    auto load8 = [&](float2 (&d)[kColHalf], int t) {
#pragma unroll
        for (int j = 0; j < kColHalf; ++j) {
            if constexpr (FULL) d[j] = buf_ld_aux<kAnaLoad>(rx, vx, 8u * ((unsigned)(t + j) << lgM));
            else d[j] = (t + j < nvalid) ? xg[(unsigned)(t + j) << lgM] : make_float2(0.f, 0.f);
        }
    };
    float2 *yb = y + wg_first * M;
    // FIR over 8 frames of the column (ring slots static: run, t0 are multiples of 16 and P divides 16), then
    // the M-point transforms of all G*8 frames of the workgroup and coalesced [frame][channel] stores.  The
    // column's samples for the half tile after next are requested before the transforms start, so their HBM
    // latency is hidden behind the LDS passes.
    auto half_tile = [&](float2 (&xin)[kColHalf], int t, auto slot0) {
        constexpr int S0 = decltype(slot0)::value;
#pragma unroll
        for (int j = 0; j < kColHalf; ++j) {
            w[(S0 + j) % P] = xin[j];
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int n = 0; n < P; ++n) {
                const float2 sv = w[(S0 + j - n + 4 * P) % P];
                acc.x = fmaf(sv.x, hc[n], acc.x);
                acc.y = fmaf(sv.y, hc[n], acc.y);
            }
            va[(g * kColHalf + j) * pitch + c] = acc;
        }
        if (t + kAhead < run) load8(xin, t + kAhead);
        __syncthreads();
        stockham_pass<R0, -1, true>(va, vb, M, 1, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
        const float2 *res = vb;
        if constexpr (R1 >= 8 && M / R1 >= 16) {
            // last pass straight from registers to y: 16-lane runs of 128 contiguous bytes (M = 128, 256); no barrier
            // behind it: the next half tile writes va first, and its own barrier stands between this pass's reads of vb
            // and the next writes to it (as in firpfbch2_col_kernel)
            stockham_last_pass_out<R1, -1>(vb, M, nq, twl, 1, true, pitch, [&](int q, int k, float2 v) {
                const int gq = q / kColHalf, fr = q - gq * kColHalf;
                if constexpr (FULL)
                    buf_st_aux<kAnaStore>(ry, 8u * (((unsigned)(gq * run + fr) << lgM) + k), 8u * ((unsigned)t << lgM), v);
                else if (t + fr < group_frames(gq))
                    yb[((unsigned)(gq * run + t + fr) << lgM) + k] = v;
            });
            return;
        } else if constexpr (R1 > 1) {
            stockham_pass<R1, -1, true>(vb, va, M, R0, nq, twl, 1, true, pitch, lgnq);
            __syncthreads();
            res = va;
        }
        // transform q = g'*8 + j is frame t + j of group g'
#pragma unroll
        for (int i = 0; i < kColHalf; ++i) {
            const int e = threadIdx.x + 256 * i;
            const int q = e >> lgM, k = e & (M - 1);
            const int gq = q / kColHalf, fr = q - gq * kColHalf;
            if constexpr (FULL)
                buf_st_aux<kAnaStore>(ry, 8u * (((unsigned)(gq * run + fr) << lgM) + k), 8u * ((unsigned)t << lgM), res[q * pitch + k]);
            else if (t + fr < group_frames(gq))
                yb[((unsigned)(gq * run + t + fr) << lgM) + k] = res[q * pitch + k];
        }
        __syncthreads();
    };
    load8(xa, 0);
    if constexpr (!kOneInFlight) load8(xb, kColHalf);
    for (int t0 = 0; t0 < run; t0 += kColTile) {
        half_tile(xa, t0, std::integral_constant<int, 0>{});
        if constexpr (kOneInFlight) half_tile(xa, t0 + kColHalf, std::integral_constant<int, kColHalf>{});
        else half_tile(xb, t0 + kColHalf, std::integral_constant<int, kColHalf>{});
    }
}

template <int P, int LGM>
static int launch_firpfbch_col(const cf32 *hist, const cf32 *x, const float *h, const cf32 *twM,
                               cf32 *y, size_t nframes, hipStream_t st, cf32 *hist_next, int p_real) {
    constexpr int M = 1 << LGM;
    const int G = 256 / M;
    // run length per column group: long enough to amortise the (p-1)-frame halo, short enough for
    // >= ~2048 workgroups
    // M = 64 on long blocks: 2048 workgroups measure 6-8 % faster than 1024 (8192: 4 % slower) although each run re-reads its
    // p - 1 frames of history; with 16 taps (one half tile in flight, four resident workgroups per CU) 4096.  Shorter blocks
    // keep runs of >= 64 (128) frames -- at 2^24 samples 4096 workgroups would leave runs of 16 frames: 1.46x read traffic, +5 %
    size_t wgs = kColWgs;
    if (LGM == 6) {
        const size_t most = (P == 16 ? 4 : 2) * kColWgs, want = nframes / ((size_t)G * (P == 16 ? 64 : 128));
        wgs = want < kColWgs ? kColWgs : (want > most ? most : want);
    }
    size_t run = nframes / (wgs * G);
    run = run / kColTile * kColTile;
    if (run < (size_t)kColTile) run = kColTile;
    if (run > 256) run = 256;
    const size_t ngroups = (nframes + run - 1) / run;
    const size_t nblk = (ngroups + G - 1) / G;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const size_t lds = (2 * (size_t)G * kColHalf * col_pitch(M, G * kColHalf) + (size_t)M) * sizeof(float2);
    static bool raised = false;
    if (!raised) {
        YG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(firpfbch_col_kernel<P, LGM, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        YG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(firpfbch_col_kernel<P, LGM, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    // every workgroup holds G * run frames (block lengths that are multiples of G * run: the streaming case): the
    // check-free kernel; otherwise the checked kernel for the whole launch (a launch of its own for the partial tail would
    // run one workgroup's whole run alone on the device)
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y), *fn = reinterpret_cast<float2 *>(hist_next);
    if (nframes % ((size_t)G * run) == 0)
        firpfbch_col_kernel<P, LGM, true><<<(unsigned)nblk, 256, lds, st>>>(fh, fx, h, ftw, fy, nframes, (int)run, fn, p_real);
    else
        firpfbch_col_kernel<P, LGM, false><<<(unsigned)nblk, 256, lds, st>>>(fh, fx, h, ftw, fy, nframes, (int)run, fn, p_real);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// firpfbch, column-sliding form for wide banks (M = 512, 1024; p in {4, 8}): one workgroup per run of frames, every
// lane owns M/256 branch columns (taps and ring per column in registers), half tiles of HF = 4096/M frames so
// the two transform buffers stay at 32 KiB each; three static radix passes (512 = 8x8x8, 1024 = 16x8x8).
// The generic tiled kernel re-reads every sample (tile + p - 1)/tile times and got 0.4-0.8 TB/s on these shapes.
// ---------------------------------------------------------------------------------------------
template <int P, int LGM, bool FULL>                // FULL: every workgroup of the launch holds `run` frames (see firpfbch_col_kernel)
__global__ void __launch_bounds__(256)
firpfbch_wide_kernel(const float2 *__restrict__ hist, const float2 *__restrict__ x,
                     const float *__restrict__ h, const float2 *__restrict__ twM,
                     float2 *__restrict__ y, size_t nframes, int run, int p_real /* <= P: the other taps are zero */) {
    constexpr int M = 1 << LGM, lgM = LGM, C = M / 256, HF = 4096 / M, TILE = 2 * HF;
    constexpr int R0 = LGM == 9 ? 8 : 16;                        // then 8 x 8
    constexpr int nq = HF, lgnq = LGM == 9 ? 3 : 2;
    constexpr int pitch = M + 32 / nq;
    static_assert(TILE % P == 0, "ring slots must be static");
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *va = reinterpret_cast<float2 *>(smem);               // [HF frames][pitch]
    float2 *vb = va + nq * pitch;
    float2 *twl = vb + nq * pitch;                               // M
    const int t = threadIdx.x;
    for (int e = t; e < M; e += 256) twl[e] = twM[e];
    float hc[C][P];
#pragma unroll
    for (int cc = 0; cc < C; ++cc)
#pragma unroll
        for (int n = 0; n < P; ++n) hc[cc][n] = n < p_real ? h[(M - 1 - (t + 256 * cc)) + n * M] : 0.0f;
    const int hist_len = (p_real - 1) * M;
    const long long x_len = (long long)nframes * M;
    const long long f_begin = (long long)blockIdx.x * run;
    const long long left = (long long)nframes - f_begin;
    const int nvalid = (int)(left < run ? left : run);
    const float2 *xg = x + f_begin * M + t;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + f_begin * M, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + f_begin * M, 0xffffffffu);
    float2 w[C][P];
#pragma unroll
    for (int cc = 0; cc < C; ++cc)
#pragma unroll
        for (int n = 1; n < P; ++n)
            w[cc][P - n] = load_hist(hist, hist_len, x, (f_begin - n) * M + t + 256 * cc, x_len);
    float2 xa[C][HF], xb[C][HF];
    auto loadh = [&](float2 (&d)[C][HF], int f) {
#pragma unroll
        for (int j = 0; j < HF; ++j)
#pragma unroll
            for (int cc = 0; cc < C; ++cc)
                if constexpr (FULL) d[cc][j] = buf_ld_aux<kAnaLoad>(rx, 8u * (t + 256u * cc), 8u * ((unsigned)(f + j) << lgM));
                else d[cc][j] = (f + j < nvalid) ? xg[((unsigned)(f + j) << lgM) + 256u * cc] : make_float2(0.f, 0.f);
    };
    float2 *yb = y + f_begin * M;
    auto half_tile = [&](float2 (&xin)[C][HF], int f, auto slot0) {
        constexpr int S0 = decltype(slot0)::value;
#pragma unroll
        for (int j = 0; j < HF; ++j)
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                w[cc][(S0 + j) % P] = xin[cc][j];
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int n = 0; n < P; ++n) {
                    const float2 sv = w[cc][(S0 + j - n + 4 * P) % P];
                    acc.x = fmaf(sv.x, hc[cc][n], acc.x);
                    acc.y = fmaf(sv.y, hc[cc][n], acc.y);
                }
                va[j * pitch + t + 256 * cc] = acc;
            }
        if (f + TILE < run) loadh(xin, f + TILE);
        __syncthreads();
        stockham_pass<R0, -1, true>(va, vb, M, 1, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
        stockham_pass<8, -1, true>(vb, va, M, R0, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
        stockham_pass<8, -1, true>(va, vb, M, R0 * 8, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < HF * C; ++i) {
            const int e = t + 256 * i;
            const int q = e >> lgM, k = e & (M - 1);
            if constexpr (FULL) buf_st_aux<kAnaStore>(ry, 8u * (((unsigned)q << lgM) + k), 8u * ((unsigned)f << lgM), vb[q * pitch + k]);
            else if (f + q < nvalid) yb[((unsigned)(f + q) << lgM) + k] = vb[q * pitch + k];
        }
        __syncthreads();
    };
    loadh(xa, 0);
    loadh(xb, HF);
    for (int f0 = 0; f0 < run; f0 += TILE) {
        half_tile(xa, f0, std::integral_constant<int, 0>{});
        half_tile(xb, f0 + HF, std::integral_constant<int, HF>{});
    }
}

template <int P, int LGM>
static int launch_firpfbch_wide(const cf32 *hist, const cf32 *x, const float *h, const cf32 *twM,
                                cf32 *y, size_t nframes, hipStream_t st, int p_real) {
    constexpr int M = 1 << LGM, HF = 4096 / M, TILE = 2 * HF;
    size_t run = nframes / kColWgs;
    run = run / TILE * TILE;
    if (run < (size_t)TILE) run = TILE;
    if (run > 256) run = 256;
    const size_t nblk = (nframes + run - 1) / run;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const size_t lds = (2 * (size_t)HF * (M + 32 / HF) + (size_t)M) * sizeof(float2);
    static bool raised = false;
    if (!raised) {
        YG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(firpfbch_wide_kernel<P, LGM, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        YG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(firpfbch_wide_kernel<P, LGM, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y);
    if (nframes % run == 0)
        firpfbch_wide_kernel<P, LGM, true><<<(unsigned)nblk, 256, lds, st>>>(fh, fx, h, ftw, fy, nframes, (int)run, p_real);
    else
        firpfbch_wide_kernel<P, LGM, false><<<(unsigned)nblk, 256, lds, st>>>(fh, fx, h, ftw, fy, nframes, (int)run, p_real);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

static constexpr size_t kChanLdsBudget = 38 * 1024;   // <= 4 workgroups per CU

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

int launch_firpfbch(const cf32 *hist, const cf32 *x, const float *h, int M, int p,
                    const cf32 *twM, cf32 *y, size_t nframes, hipStream_t st, cf32 *hist_next, bool *hist_written) {
    if (hist_written) *hist_written = false;
    if (nframes == 0) return YAGI_OK;
    if ((M == 512 || M == 1024) && p <= 8 && nframes >= 64) {     // p < 8 other than 4: zero taps behind the real ones
        if (M == 512) return p <= 4 ? launch_firpfbch_wide<4, 9>(hist, x, h, twM, y, nframes, st, p)
                                    : launch_firpfbch_wide<8, 9>(hist, x, h, twM, y, nframes, st, p);
        return p <= 4 ? launch_firpfbch_wide<4, 10>(hist, x, h, twM, y, nframes, st, p)
                      : launch_firpfbch_wide<8, 10>(hist, x, h, twM, y, nframes, st, p);
    }
    if ((M == 8 || M == 16 || M == 32 || M == 64 || M == 128 || M == 256) && nframes >= 64) {
#define YG_COL_CASE(PP)                                                                              \
    case PP:                                                                                         \
        if (hist_written) *hist_written = hist_next != nullptr;                                      \
        return M == 8 ? launch_firpfbch_col<PP, 3>(hist, x, h, twM, y, nframes, st, hist_next, p)                  \
             : M == 16 ? launch_firpfbch_col<PP, 4>(hist, x, h, twM, y, nframes, st, hist_next, p)                 \
             : M == 32 ? launch_firpfbch_col<PP, 5>(hist, x, h, twM, y, nframes, st, hist_next, p)                 \
             : M == 64 ? launch_firpfbch_col<PP, 6>(hist, x, h, twM, y, nframes, st, hist_next, p)                 \
             : M == 128 ? launch_firpfbch_col<PP, 7>(hist, x, h, twM, y, nframes, st, hist_next, p)                \
                        : launch_firpfbch_col<PP, 8>(hist, x, h, twM, y, nframes, st, hist_next, p);
        // branch lengths between the built sizes take the next one with zero taps behind theirs
        switch (p <= 4 ? 4 : p <= 8 ? 8 : p <= 16 ? 16 : 0) {
            YG_COL_CASE(4)
            YG_COL_CASE(8)
            YG_COL_CASE(16)
            default: break;
        }
#undef YG_COL_CASE
    }
    // frames per tile: as many as fit the LDS budget (>= 1)
    int F = 4096 / M;
    if (F < 1) F = 1;
    const size_t fixed = (size_t)M * sizeof(float2) + (size_t)p * M * sizeof(float);
    auto need = [&](int f) { return ((size_t)(f + p - 1) * M + 2 * (size_t)f * M) * sizeof(float2) + fixed; };
    // wide banks: a tile of one frame re-reads every sample p times; let such shapes use half the LDS (2 WGs per CU)
    const size_t budget = M >= 512 ? 78 * 1024 : kChanLdsBudget;
    while (F > 1 && need(F) > budget) F /= 2;
    if (need(F) > 150 * 1024) return fail(YAGI_ERR_CONFIG, "firpfbch: M*p too large for LDS (%d x %d)", M, p);
    const bool pow2 = is_pow2(M);
    const void *fn = pow2 ? reinterpret_cast<const void *>(firpfbch_kernel<true>)
                          : reinterpret_cast<const void *>(firpfbch_kernel<false>);
    if (need(F) > 64 * 1024) YG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t tiles = (nframes + F - 1) / F;
    const unsigned grid = (unsigned)(tiles < 65536 ? tiles : 65536);
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y);
    if (pow2)
        firpfbch_kernel<true><<<grid, 256, need(F), st>>>(fh, fx, h, M, p, ftw, FacList{0, {0}}, make_pow2_plan(M), fy, nframes, F);
    else
        firpfbch_kernel<false><<<grid, 256, need(F), st>>>(fh, fx, h, M, p, ftw, factorize_small(M), Pow2Plan{0, {0}}, fy, nframes, F);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// firpfbch synthesizer (liquid-dsp firpfbch_crcf_synthesizer_execute semantics; absent from the reference):
//   v_f[i] = sum_k X_f[k] e^{+j 2 pi i k / M}                       (unnormalised inverse DFT of frame f)
//   y[f M + i] = sum_{n < p} h[i + n M] v_{f-n}[i]                  (branch i: window push, then dot product)
// One workgroup = a tile of F output frames: it loads and inverse-transforms frames f0-(p-1) .. f0+F-1 (the first
// p-1 of them from the previous tile / the history) in LDS and then evaluates the F M branch dot products with
// the taps read coalesced from L2.  POW2 as in firpfbch_kernel.
// ---------------------------------------------------------------------------------------------
template <bool POW2>
__global__ void __launch_bounds__(256)
firpfbch_syn_kernel(const float2 *__restrict__ hist, const float2 *__restrict__ x,
                    const float *__restrict__ h, int M, int p, const float2 *__restrict__ twM,
                    FacList fl, Pow2Plan plan, float2 *__restrict__ y, size_t nframes, int F) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *va = reinterpret_cast<float2 *>(smem);              // (F + p - 1) * M
    float2 *vb = va + (size_t)(F + p - 1) * M;                  // (F + p - 1) * M
    float2 *twl = vb + (size_t)(F + p - 1) * M;                 // M
    const int hist_len = (p - 1) * M;
    for (int e = threadIdx.x; e < M; e += 256) twl[e] = twM[e];
    const size_t f0 = (size_t)blockIdx.x * F;
    const int nf = (int)((nframes - f0) < (size_t)F ? (nframes - f0) : (size_t)F);
    const long long base = (long long)f0 * M - hist_len;
    const int ntr = nf + p - 1, nspan = ntr * M;
    const long long x_len = (long long)nframes * M;
    if (base >= 0 && base + nspan <= x_len) {
        const float2 *src = x + base;
        for (int u = threadIdx.x; u < nspan; u += 256) va[u] = src[u];
    } else {
        for (int u = threadIdx.x; u < nspan; u += 256) va[u] = load_hist(hist, hist_len, x, base + u, x_len);
    }
    __syncthreads();
    const float2 *res = POW2 ? lds_fft_pow2<+1>(va, vb, M, ntr, plan, twl, 1, true)
                             : lds_dft_frames(va, vb, M, ntr, fl, twl, 1, true);
    for (int e = threadIdx.x; e < nf * M; e += 256) {
        const int f = e / M, i = e - f * M;
        float2 acc = make_float2(0.f, 0.f);
        const float2 *vp = res + (f + p - 1) * M + i;            // v_{f0+f}[i]; v_{f0+f-n}[i] is n*M before it
        for (int n = 0; n < p; ++n) {
            const float hv = h[i + n * M];
            const float2 sv = vp[-n * M];
            acc.x = fmaf(sv.x, hv, acc.x);
            acc.y = fmaf(sv.y, hv, acc.y);
        }
        y[f0 * M + e] = acc;
    }
}

// Column-sliding synthesizer (M in {8..256}, p in {4,8,16}): the mirror image of firpfbch_col_kernel.  Per half tile of
// 8 frames: coalesced loads of the channel frames -> LDS -> static inverse radix passes -> every lane (one branch
// column i) takes its 8 values back, pushes them through its p-deep register ring and stores y[f M + i] coalesced.
// A run starts with W = 8 (p <= 8) or 16 warm-up frames (the p-1 frames before it, inverse-transformed again)
// whose outputs are not stored.
// FAST: all runs of the workgroup are full: channel frames and outputs go through buffer descriptors (one VGPR offset per
// lane, frame steps in SGPRs), no range checks; only workgroup 0's warm-up frames take the history path.  A launch with a
// partial last workgroup takes FAST = false as a whole.
template <int P, int LGM, bool FAST>
__global__ void __launch_bounds__(256)
firpfbch_syn_col_kernel(const float2 *__restrict__ hist, const float2 *__restrict__ x,
                        const float *__restrict__ h, const float2 *__restrict__ twM,
                        float2 *__restrict__ y, size_t nframes, int run, int p_real /* <= P: the other taps are zero */) {
    constexpr int M = 1 << LGM, lgM = LGM;
    constexpr int R0 = (LGM == 3 || LGM == 5 || LGM == 6) ? 8 : 16, R1 = M / R0;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int G = 256 / M;
    constexpr int nq = G * kColHalf, lgnq = 8 - LGM + 3;
    constexpr int pitch = col_pitch(M, nq);
    constexpr int W = P <= 8 ? 8 : 16;                           // warm-up frames per run (>= P - 1)
    float2 *va = reinterpret_cast<float2 *>(smem);
    float2 *vb = va + nq * pitch;
    float2 *twl = vb + nq * pitch;
    const int g = threadIdx.x >> lgM, c = threadIdx.x & (M - 1);
    for (int e = threadIdx.x; e < M; e += 256) twl[e] = twM[e];
    float hc[P];
#pragma unroll
    for (int n = 0; n < P; ++n) hc[n] = n < p_real ? h[c + n * M] : 0.0f;
    const int hist_len = (p_real - 1) * M;
    const long long x_len = (long long)nframes * M;
    const long long wg_first = (long long)blockIdx.x * G * run;
    const long long f_begin = wg_first + (long long)g * run;
    const long long left = (long long)nframes - f_begin;
    const int nvalid = (int)(left < 0 ? 0 : (left < run ? left : run));
    const bool first_wg = blockIdx.x == 0;                 // its warm-up frames come from the history
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(FAST && !first_wg ? x + (wg_first - W) * M : x, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(FAST ? y + wg_first * M : y, 0xffffffffu);
    const unsigned vx = 8u * ((unsigned)(g * run) * M + c);
    const unsigned sx0 = first_wg ? 0u : 8u * ((unsigned)W << lgM);             // descriptor offset of frame t = 0
    float2 w[P];
#pragma unroll
    for (int n = 0; n < P; ++n) w[n] = make_float2(0.f, 0.f);
    auto half_tile = [&](int t, auto slot0) {
        constexpr int S0 = decltype(slot0)::value;
        // channel frames t .. t+7 of this group (negative: the frames before the run; beyond the end: zeros)
#pragma unroll
        for (int j = 0; j < kColHalf; ++j) {
            if (FAST && (t >= 0 || !first_wg)) {                 // uniform: only workgroup 0's warm-up needs the history
                va[(g * kColHalf + j) * pitch + c] = buf_ld(rx, vx, sx0 + 8u * (unsigned)((t + j) << lgM));
            } else {
                const long long f = f_begin + t + j;
                va[(g * kColHalf + j) * pitch + c] = (t + j < nvalid) ? load_hist(hist, hist_len, x, f * M + c, x_len)
                                                                     : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        stockham_pass<R0, +1, true>(va, vb, M, 1, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
        const float2 *res = vb;
        if constexpr (R1 > 1) {
            stockham_pass<R1, +1, true>(vb, va, M, R0, nq, twl, 1, true, pitch, lgnq);
            __syncthreads();
            res = va;
        }
        // two frames at a time as one 4-wide FMA per lag (a dependent packed FMA directly behind its producer costs a
        // wait state); frame j's oldest sample shares its ring slot with frame j + 1's newest, so it is kept aside; every
        // frame still adds its taps in lag order
        // (P = 16: the paired form needs 129-154 VGPRs, 3 waves per SIMD: one frame at a time)
        typedef float v4f_t __attribute__((ext_vector_type(4)));
        if constexpr (P > 8) {
#pragma unroll
            for (int j = 0; j < kColHalf; ++j) {
                w[(S0 + j) % P] = res[(g * kColHalf + j) * pitch + c];
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int n = 0; n < P; ++n) {
                    const float2 sv = w[(S0 + j - n + 4 * P) % P];
                    acc.x = fmaf(sv.x, hc[n], acc.x);
                    acc.y = fmaf(sv.y, hc[n], acc.y);
                }
                if constexpr (FAST) {
                    if (t >= 0) buf_st(ry, vx, 8u * ((unsigned)(t + j) << lgM), acc);   // t: uniform, a multiple of 8
                } else if (t + j >= 0 && t + j < nvalid) y[(f_begin + t + j) * M + c] = acc;
            }
        } else
#pragma unroll
        for (int j = 0; j < kColHalf; j += 2) {
            w[(S0 + j) % P] = res[(g * kColHalf + j) * pitch + c];
            const float2 oldest = w[(S0 + j + 1) % P];           // lag P - 1 of frame j
            w[(S0 + j + 1) % P] = res[(g * kColHalf + j + 1) * pitch + c];
            v4f_t a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < P; ++n) {
                const float2 s0 = n == P - 1 ? oldest : w[(S0 + j - n + 4 * P) % P];
                const float2 s1 = w[(S0 + j + 1 - n + 4 * P) % P];
                a2 = __builtin_elementwise_fma(v4f_t{s0.x, s0.y, s1.x, s1.y}, v4f_t{hc[n], hc[n], hc[n], hc[n]}, a2);
            }
            const float2 acc0 = make_float2(a2.x, a2.y), acc1 = make_float2(a2.z, a2.w);
            if constexpr (FAST) {
                if (t >= 0) {                                    // t: uniform, a multiple of 8
                    buf_st(ry, vx, 8u * ((unsigned)(t + j) << lgM), acc0);
                    buf_st(ry, vx, 8u * ((unsigned)(t + j + 1) << lgM), acc1);
                }
            } else {
                if (t + j >= 0 && t + j < nvalid) y[(f_begin + t + j) * M + c] = acc0;
                if (t + j + 1 >= 0 && t + j + 1 < nvalid) y[(f_begin + t + j + 1) * M + c] = acc1;
            }
        }
        __syncthreads();                             // the next half tile overwrites va / vb
    };
    if constexpr (P <= 8) {
        for (int t0 = -W; t0 < run; t0 += kColHalf) half_tile(t0, std::integral_constant<int, 0>{});
    } else {
        for (int t0 = -W; t0 < run; t0 += kColTile) {
            half_tile(t0, std::integral_constant<int, 0>{});
            half_tile(t0 + kColHalf, std::integral_constant<int, kColHalf>{});
        }
    }
}

template <int P, int LGM>
static int launch_firpfbch_syn_col(const cf32 *hist, const cf32 *x, const float *h, const cf32 *twM,
                                   cf32 *y, size_t nframes, hipStream_t st, int p_real) {
    constexpr int M = 1 << LGM;
    const int G = 256 / M;
    size_t run = nframes / (kColWgs * G);
    run = run / kColTile * kColTile;
    if (run < (size_t)kColTile) run = kColTile;
    if (run > 256) run = 256;
    const size_t ngroups = (nframes + run - 1) / run;
    const size_t nblk = (ngroups + G - 1) / G;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const size_t lds = (2 * (size_t)G * kColHalf * col_pitch(M, G * kColHalf) + (size_t)M) * sizeof(float2);
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y);
    if (nframes % ((size_t)G * run) == 0)            // every workgroup full (see launch_firpfbch_col)
        firpfbch_syn_col_kernel<P, LGM, true><<<(unsigned)nblk, 256, lds, st>>>(fh, fx, h, ftw, fy, nframes, (int)run, p_real);
    else
        firpfbch_syn_col_kernel<P, LGM, false><<<(unsigned)nblk, 256, lds, st>>>(fh, fx, h, ftw, fy, nframes, (int)run, p_real);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

int launch_firpfbch_syn(const cf32 *hist, const cf32 *x, const float *h, int M, int p,
                        const cf32 *twM, cf32 *y, size_t nframes, hipStream_t st) {
    if (nframes == 0) return YAGI_OK;
    if ((M == 8 || M == 16 || M == 32 || M == 64 || M == 128 || M == 256) && nframes >= 64) {
#define YG_SYN_CASE(PP)                                                                              \
    case PP:                                                                                         \
        return M == 8 ? launch_firpfbch_syn_col<PP, 3>(hist, x, h, twM, y, nframes, st, p)              \
             : M == 16 ? launch_firpfbch_syn_col<PP, 4>(hist, x, h, twM, y, nframes, st, p)             \
             : M == 32 ? launch_firpfbch_syn_col<PP, 5>(hist, x, h, twM, y, nframes, st, p)             \
             : M == 64 ? launch_firpfbch_syn_col<PP, 6>(hist, x, h, twM, y, nframes, st, p)             \
             : M == 128 ? launch_firpfbch_syn_col<PP, 7>(hist, x, h, twM, y, nframes, st, p)            \
                        : launch_firpfbch_syn_col<PP, 8>(hist, x, h, twM, y, nframes, st, p);
        switch (p <= 4 ? 4 : p <= 8 ? 8 : p <= 16 ? 16 : 0) {   // zero-padded branch lengths (see launch_firpfbch)
            YG_SYN_CASE(4)
            YG_SYN_CASE(8)
            YG_SYN_CASE(16)
            default: break;
        }
#undef YG_SYN_CASE
    }
    int F = 4096 / M;
    if (F < 1) F = 1;
    auto need = [&](int f) { return (2 * (size_t)(f + p - 1) * M + (size_t)M) * sizeof(float2); };
    const size_t budget = 78 * 1024;
    while (F > 1 && need(F) > budget) F /= 2;
    if (need(F) > 150 * 1024) return fail(YAGI_ERR_CONFIG, "firpfbch synthesizer: M*p too large for LDS (%d x %d)", M, p);
    const bool pow2 = is_pow2(M);
    const void *fn = pow2 ? reinterpret_cast<const void *>(firpfbch_syn_kernel<true>)
                          : reinterpret_cast<const void *>(firpfbch_syn_kernel<false>);
    if (need(F) > 64 * 1024) YG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t tiles = (nframes + F - 1) / F;
    if (tiles > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y);
    if (pow2)
        firpfbch_syn_kernel<true><<<(unsigned)tiles, 256, need(F), st>>>(fh, fx, h, M, p, ftw, FacList{0, {0}}, make_pow2_plan(M), fy, nframes, F);
    else
        firpfbch_syn_kernel<false><<<(unsigned)tiles, 256, need(F), st>>>(fh, fx, h, M, p, ftw, factorize_small(M), Pow2Plan{0, {0}}, fy, nframes, F);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// firpfbch2 analyzer (optionally one rank's sub-band shard)
// ---------------------------------------------------------------------------------------------
template <bool POW2>
__global__ void __launch_bounds__(256)
firpfbch2_kernel(const float2 *__restrict__ hist, int hist_len, const float2 *__restrict__ x,
                 const float *__restrict__ h, int M, int p, const float2 *__restrict__ twM,
                 FacList fl, Pow2Plan plan, unsigned long long step0, int rank, int R,
                 float2 *__restrict__ y, size_t nsteps, int S) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int M2 = M / 2, Mr = M / R;
    float2 *va = reinterpret_cast<float2 *>(smem);             // S * M
    float2 *vb = va + (size_t)S * M;                           // S * M (fold / Stockham partner)
    float2 *twl = vb + (size_t)S * M;                          // M
    float *hs = reinterpret_cast<float *>(twl + M);            // p * M taps, natural order h[i + n*M]
    const float invM = 1.0f / (float)M;
    const long long x_len = (long long)nsteps * M2;
    for (int e = threadIdx.x; e < M; e += 256) twl[e] = twM[e];
    for (int e = threadIdx.x; e < p * M; e += 256) hs[e] = h[e];
    __syncthreads();
    for (size_t tile = blockIdx.x; tile * S < nsteps; tile += gridDim.x) {
        const size_t s0 = tile * S;
        const int ns = (int)((nsteps - s0) < (size_t)S ? (nsteps - s0) : (size_t)S);
        // oldest sample any window of this tile needs: (s0 - 1)*M2 - (p-1)*M  (>= 0 for interior tiles)
        const bool interior = (long long)s0 * M2 - M2 - (long long)(p - 1) * M >= 0;
        for (int e = threadIdx.x; e < ns * M; e += 256) {
            const int sl = e / M, b = e - sl * M;
            const unsigned long long sg = step0 + s0 + sl;     // global step index (parity = flag)
            const int flag = (int)(sg & 1ull);
            const int bpar = (b >= M2) ? 1 : 0;
            const int pos = bpar ? (M - 1 - b) : (M2 - 1 - b);
            const int back = (flag == bpar) ? 0 : 1;           // steps since window b was last fed
            int i = b - (flag ? M2 : 0);
            if (i < 0) i += M;
            // newest sample of window b: X index (s0 + sl - back)*M2 + pos; lanes hold consecutive b,
            // i.e. a contiguous (descending) run of x; re-reads by later steps hit L1/L2
            const long long top = ((long long)(s0 + sl) - back) * M2 + pos;
            const float *hp = hs + i;
            float2 acc = make_float2(0.f, 0.f);
            if (interior) {
                const float2 *xp = x + top;
                for (int n = 0; n < p; ++n) {
                    const float hv = hp[n * M];
                    const float2 sv = xp[-(long long)n * M];
                    acc.x = fmaf(sv.x, hv, acc.x);
                    acc.y = fmaf(sv.y, hv, acc.y);
                }
            } else {
                for (int n = 0; n < p; ++n) {
                    const float hv = hp[n * M];
                    const float2 sv = load_hist(hist, hist_len, x, top - (long long)n * M, x_len);
                    acc.x = fmaf(sv.x, hv, acc.x);
                    acc.y = fmaf(sv.y, hv, acc.y);
                }
            }
            va[e] = acc;
        }
        __syncthreads();
        float2 *srcb = va, *dstb = vb;
        if (R > 1) {
            // fold to the rank's residue class: Z[b'] = W_M^{-b' r} sum_a W_R^{-a r} V[Mr*a + b']
            for (int e = threadIdx.x; e < ns * Mr; e += 256) {
                const int sl = e / Mr, bq = e - sl * Mr;
                float2 acc = make_float2(0.f, 0.f);
                for (int a = 0; a < R; ++a) {
                    float2 w = twl[((a * rank) % R) * Mr];     // W_R^{a r} = W_M^{a r M/R}
                    w.y = -w.y;
                    acc = cadd(acc, cmul(va[sl * M + Mr * a + bq], w));
                }
                float2 w2 = twl[(bq * rank) % M];
                w2.y = -w2.y;
                vb[e] = cmul(acc, w2);
            }
            __syncthreads();
            srcb = vb;
            dstb = va;
        }
        float2 *res = POW2 ? lds_fft_pow2<+1>(srcb, dstb, Mr, ns, plan, twl, R)
                           : lds_dft_frames(srcb, dstb, Mr, ns, fl, twl, R, true);
        for (int e = threadIdx.x; e < ns * Mr; e += 256) {
            const float2 v = res[e];
            y[s0 * Mr + e] = make_float2(v.x * invM, v.y * invM);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// firpfbch2, column-sliding form (M in {64,128,256}, branch length P = 2m in {2,4,8}, even first step):
// lane b owns window b.  A window is fed once per PAIR of steps (even steps feed b < M/2, odd steps feed
// b >= M/2) and produces an output on both steps of the pair with two different tap sets
// (i = b on even steps, i = b - M/2 mod M on odd steps).  Per pair and lane: one coalesced 8-byte load and
// 2*P packed FMAs out of registers.  To keep the code free of divergence the even-step taps of the lanes
// with b >= M/2 (which must see the window BEFORE this pair's sample) are stored rotated by one slot.
// ---------------------------------------------------------------------------------------------
// M = 2^LGM is a template parameter (see firpfbch_col_kernel); SHARDED = a rank's sub-band shard (R > 1: fold to the
// residue class, then a run-time planned M/R-point transform), otherwise two static radix passes.
// FULL: every workgroup of the launch holds G * run steps -- samples and outputs go through buffer descriptors (one
// VGPR byte offset per lane, the steps in SGPRs / immediates: no 64-bit address arithmetic, no per-step checks); a
// launch with a partial last workgroup takes FULL = false as a whole.
template <int P, int LGM, bool SHARDED, bool FULL>
__global__ void __launch_bounds__(256)
firpfbch2_col_kernel(const float2 *__restrict__ hist, int hist_len, const float2 *__restrict__ x,
                     const float *__restrict__ h, const float2 *__restrict__ twM, Pow2Plan plan,
                     int rank, int R, float2 *__restrict__ y, size_t nsteps, int run /* steps, multiple of 16 */,
                     float2 *__restrict__ hist_next, int p_real /* taps per branch, <= P: the rest are zero */) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int M = 1 << LGM, lgM = LGM, G = 256 / M, M2 = M / 2;
    chan_write_next_hist(hist, hist_len, x, nsteps * (size_t)M2, hist_next);
    constexpr int R0 = (LGM == 3 || LGM == 5 || LGM == 6) ? 8 : 16, R1 = M / R0;   // 8, 16: one pass; 32 = 8 x 4
    constexpr int nq = G * kColHalf, lgnq = 8 - LGM + 3;       // transforms in flight (kColHalf = 8)
    constexpr int pitch = col_pitch(M, nq);
    const int Mr = SHARDED ? M / R : M;
    const int lgMr = 31 - __builtin_clz((unsigned)Mr);
    constexpr bool kRegTile = !SHARDED && LGM == 8;             // see the end of the kernel
    constexpr int kRegPitch = 272;                              // = 16 mod 32: two transforms of a half wave on disjoint banks
    float2 *va = reinterpret_cast<float2 *>(smem);              // [256/M groups][8 steps][pitch]; kRegTile: [16 steps][272]
    float2 *vb = va + nq * pitch;
    float2 *twl = vb + nq * pitch;                              // M
    const int g = threadIdx.x >> lgM, b = threadIdx.x & (M - 1);
    if constexpr (!kRegTile)
        for (int e = threadIdx.x; e < M; e += 256) twl[e] = twM[e];
    const int bpar = (b >= M2) ? 1 : 0;
    const int pos = bpar ? (M - 1 - b) : (M2 - 1 - b);
    const int i1 = (b - M2 + M) & (M - 1);
    float h0r[P], h1[P];                                         // h0r[m] pairs with ring slot (kk - m)
#pragma unroll
    for (int m = 0; m < P; ++m) {
        const int n0 = bpar ? (m + P - 1) % P : m;              // rotated for the late-fed half
        h0r[m] = n0 < p_real ? h[b + n0 * M] : 0.0f;            // branch lengths between the built sizes run zero-padded
        h1[m] = m < p_real ? h[i1 + m * M] : 0.0f;
    }
    const float invM = 1.0f / (float)M;
    const long long x_len = (long long)nsteps * M2;
    // one workgroup = G consecutive runs of `run` steps; step indices below are 32-bit offsets from its first
    const long long wg_first = (long long)blockIdx.x * G * run;
    const long long wg_left = (long long)nsteps - wg_first;
    const int wg_steps = (int)(wg_left < (long long)G * run ? wg_left : (long long)G * run);
    auto group_steps = [&](int gq) {
        const int v = wg_steps - gq * run;
        return v < 0 ? 0 : (v > run ? run : v);
    };
    const long long s_begin = wg_first + (long long)g * run;             // even
    const int nvalid = group_steps(g);
    // sample of local pair kk: X[(s_begin + 2kk + bpar)*M2 + pos], kept in ring slot kk mod P
    const float2 *xg = x + (s_begin + bpar) * M2 + pos;                   // + 2kk*M2 = kk*M
    float2 w[P];
#pragma unroll
    for (int n = 1; n <= P; ++n)                                         // pairs -1 .. -P
        w[(P - n) % P] = load_hist(hist, hist_len, x, (s_begin - 2 * n + bpar) * M2 + pos, x_len);
    constexpr int kPairs = kColHalf / 2;                                 // 4 samples feed 8 steps
    float2 xa[kPairs], xb[kPairs];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + wg_first * M2, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + wg_first * Mr, 0xffffffffu);
    const unsigned vx = 8u * ((unsigned)(g * run + bpar) * M2 + pos);
    auto load4 = [&](float2 (&d)[kPairs], int t /* first step of the half tile */) {
#pragma unroll
        for (int kk = 0; kk < kPairs; ++kk) {
            if constexpr (FULL) d[kk] = buf_ld_aux<kAnaLoad>(rx, vx, 8u * ((unsigned)(t / 2 + kk) << lgM));
            else d[kk] = (t + 2 * kk + bpar < nvalid) ? xg[(unsigned)(t / 2 + kk) << lgM] : make_float2(0.f, 0.f);
        }
    };
    float2 *yb = y + wg_first * Mr;
    auto half_tile = [&](float2 (&xin)[kPairs], int t, auto slot0) {
        constexpr int S0 = decltype(slot0)::value;                       // ring slot of the half tile's first pair
#pragma unroll
        for (int kk = 0; kk < kPairs; ++kk) {
            const float2 old = w[(S0 + kk) % P];
            // even step: the early-fed half already sees the new sample, the late-fed half the old one; odd step:
            // everybody is fed.  The two chains differ in the newest slot only; they advance together as one 4-wide
            // FMA per lag (two packed FMAs on different accumulators: a dependent packed FMA directly behind its producer
            // costs a wait state), each still adding its taps in lag order.
            typedef float v4f_t __attribute__((ext_vector_type(4)));
            const float2 new0 = bpar ? old : xin[kk];
            w[(S0 + kk) % P] = xin[kk];
            v4f_t a01 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < P; ++m) {
                const float2 s1 = w[(S0 + kk - m + 4 * P) % P];
                const float2 s0 = m == 0 ? new0 : s1;
                a01 = __builtin_elementwise_fma(v4f_t{s0.x, s0.y, s1.x, s1.y}, v4f_t{h0r[m], h0r[m], h1[m], h1[m]}, a01);
            }
            const float2 a0 = make_float2(a01.x, a01.y), a1 = make_float2(a01.z, a01.w);
            va[(g * kColHalf + 2 * kk) * pitch + b] = a0;
            va[(g * kColHalf + 2 * kk + 1) * pitch + b] = a1;
        }
        if (t + kColTile < run) load4(xin, t + kColTile);                 // in flight during the transforms
        __syncthreads();
        const float2 *res;
        if constexpr (SHARDED) {
            // fold to the rank's residue class: Z[b'] = W_M^{-b' r} sum_a W_R^{-a r} V[Mr*a + b']
            for (int e = threadIdx.x; e < nq * Mr; e += 256) {
                const int q = e >> lgMr, bq = e & (Mr - 1);
                float2 a = make_float2(0.f, 0.f);
                for (int aa = 0; aa < R; ++aa) {
                    float2 wv = twl[((aa * rank) % R) * Mr];
                    wv.y = -wv.y;
                    a = cadd(a, cmul(va[q * pitch + Mr * aa + bq], wv));
                }
                float2 w2 = twl[(bq * rank) & (M - 1)];
                w2.y = -w2.y;
                vb[q * pitch + bq] = cmul(a, w2);
            }
            __syncthreads();
            res = lds_fft_pow2<+1, true>(vb, va, Mr, nq, plan, twl, R, true, pitch, lgnq);
        } else {
            stockham_pass<R0, +1, true>(va, vb, M, 1, nq, twl, 1, true, pitch, lgnq);
            __syncthreads();
            res = vb;
            if constexpr (R1 >= 8 && M / R1 >= 16) {
                // last pass straight from registers to y: 16-lane runs of 128 contiguous bytes (M = 256)
                stockham_last_pass_out<R1, +1>(vb, M, nq, twl, 1, true, pitch, [&](int q, int k, float2 v) {
                    const int gq = q / kColHalf, sl = q - gq * kColHalf;
                    if constexpr (FULL)
                        buf_st_aux<kAnaStore>(ry, 8u * (((unsigned)(gq * run + sl) << lgM) + k), 8u * ((unsigned)t << lgM),
                               make_float2(v.x * invM, v.y * invM));
                    else if (t + sl < group_steps(gq))
                        yb[(size_t)(gq * run + t + sl) * M + k] = make_float2(v.x * invM, v.y * invM);
                });
                return;                          // no barrier: the next half tile writes va first, and its own
                                                 // barrier stands between this pass's reads of vb and the next writes
            } else if constexpr (R1 > 1) {
                stockham_pass<R1, +1, true>(vb, va, M, R0, nq, twl, 1, true, pitch, lgnq);
                __syncthreads();
                res = va;
            }
        }
        for (int e = threadIdx.x; e < nq * Mr; e += 256) {
            const int q = e >> lgMr, k = e & (Mr - 1);
            const int gq = q / kColHalf, sl = q - gq * kColHalf;
            if constexpr (FULL) {
                const float2 v = res[q * pitch + k];
                buf_st_aux<kAnaStore>(ry, 8u * ((unsigned)(gq * run + sl) * Mr + k), 8u * ((unsigned)t * Mr),
                       make_float2(v.x * invM, v.y * invM));
            } else if (t + sl < group_steps(gq)) {
                const float2 v = res[q * pitch + k];
                yb[(size_t)(gq * run + t + sl) * Mr + k] = make_float2(v.x * invM, v.y * invM);
            }
        }
        __syncthreads();
    };
    if constexpr (kRegTile) {
        // M = 256: 16 steps per tile, their 256-point transforms in registers (the scheme of fft_n256m_passes_to_regs,
        // M = 1): lane (tr, u) = (t / 16, t % 16) takes V[step tr][16 a + u] out of the tile, and the exchange between
        // the two radix-16 passes stays inside the transform's 16 lanes -- a quarter of a wave, ordered by wave
        // barriers.  Every lane is busy in both passes (the LDS Stockham form kept 128 of 256), three workgroup
        // barriers per 16 steps instead of six, and the bins leave from registers as 128-byte runs.
        const unsigned tr = threadIdx.x >> 4, u = threadIdx.x & 15u;
        float2 wq[16];                                                   // W_256^{-u c}: constant over the launch
#pragma unroll
        for (int c = 1; c < 16; ++c) {
            const float2 e = twM[(u * (unsigned)c) & 255u];
            wq[c] = make_float2(e.x, -e.y);
        }
        constexpr int kTilePairs = kColTile / 2;
        float2 xin[kTilePairs];
        auto load8 = [&](int t) {
#pragma unroll
            for (int kk = 0; kk < kTilePairs; ++kk) {
                if constexpr (FULL) xin[kk] = buf_ld_aux<kAnaLoad>(rx, vx, 8u * ((unsigned)(t / 2 + kk) << lgM));
                else xin[kk] = (t + 2 * kk + bpar < nvalid) ? xg[(unsigned)(t / 2 + kk) << lgM] : make_float2(0.f, 0.f);
            }
        };
        load8(0);
        for (int t = 0; t < run; t += kColTile) {
#pragma unroll
            for (int kk = 0; kk < kTilePairs; ++kk) {                    // the same chains as half_tile's
                typedef float v4f_t __attribute__((ext_vector_type(4)));
                const float2 old = w[kk % P];
                const float2 new0 = bpar ? old : xin[kk];
                w[kk % P] = xin[kk];
                v4f_t a01 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < P; ++m) {
                    const float2 s1 = w[(kk - m + 4 * P) % P];
                    const float2 s0 = m == 0 ? new0 : s1;
                    a01 = __builtin_elementwise_fma(v4f_t{s0.x, s0.y, s1.x, s1.y}, v4f_t{h0r[m], h0r[m], h1[m], h1[m]}, a01);
                }
                va[(2 * kk) * kRegPitch + b] = make_float2(a01.x, a01.y);
                va[(2 * kk + 1) * kRegPitch + b] = make_float2(a01.z, a01.w);
            }
            if (t + kColTile < run) load8(t + kColTile);                 // in flight during the transforms
            __syncthreads();
            float2 v[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = va[tr * kRegPitch + 16 * a + u];
            __syncthreads();                                             // the tile is out before its slots turn into exchanges
            dft16<+1>(v);
            float2 *ex = va + tr * kRegPitch;                            // [c][b], row stride 17
            ex[u] = v[dft16_pos(0)];
#pragma unroll
            for (int c = 1; c < 16; ++c) ex[c * 17 + u] = cmul(v[dft16_pos(c)], wq[c]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = ex[u * 17 + a];
            dft16<+1>(v);                                                // lane u: bins u + 16 c'
#pragma unroll
            for (int cp = 0; cp < 16; ++cp) {
                const float2 o = v[dft16_pos(cp)];
                if constexpr (FULL)
                    buf_st_aux<kAnaStore>(ry, 8u * ((tr << lgM) + u + 16u * cp), 8u * ((unsigned)t << lgM),
                                          make_float2(o.x * invM, o.y * invM));
                else if (t + (int)tr < nvalid)
                    yb[(size_t)(t + tr) * M + u + 16 * cp] = make_float2(o.x * invM, o.y * invM);
            }
            __syncthreads();                                             // exchanges read before the next tile lands
        }
        return;
    }
    load4(xa, 0);
    load4(xb, kColHalf);
    for (int t0 = 0; t0 < run; t0 += kColTile) {
        half_tile(xa, t0, std::integral_constant<int, 0>{});
        half_tile(xb, t0 + kColHalf, std::integral_constant<int, kPairs>{});
    }
}

template <int P, int LGM>
static int launch_firpfbch2_col(const cf32 *hist, int hist_len, const cf32 *x, const float *h,
                                const cf32 *twM, int rank, int nranks, cf32 *y, size_t nsteps, hipStream_t st,
                                cf32 *hist_next, int p_real) {
    constexpr int M = 1 << LGM;
    const int G = 256 / M;
    // M = 256 on long blocks: 4096 workgroups measure 3.5-4 % faster than 1024 (8192 the same); shorter blocks keep runs of
    // >= 128 steps (at 2^24 samples 4096 workgroups leave runs of 32 steps: 1.47x read traffic, 99.8 us against 85.1 us)
    size_t wgs = kColWgs;
    if (LGM == 8) {
        const size_t want = nsteps / ((size_t)G * 128);
        wgs = want < kColWgs ? kColWgs : (want > 4 * kColWgs ? 4 * kColWgs : want);
    }
    size_t run = nsteps / (wgs * G);
    run = run / kColTile * kColTile;
    if (run < (size_t)kColTile) run = kColTile;
    if (run > 512) run = 512;
    const size_t ngroups = (nsteps + run - 1) / run;
    const size_t nblk = (ngroups + G - 1) / G;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const size_t lds = (LGM == 8 && nranks <= 1)
                           ? (size_t)kColTile * 272 * sizeof(float2)       // the register-transform tile (kRegTile)
                           : (2 * (size_t)G * kColHalf * col_pitch(M, G * kColHalf) + (size_t)M) * sizeof(float2);
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y), *fn = reinterpret_cast<float2 *>(hist_next);
    // every workgroup full: the check-free kernel; otherwise the checked kernel for the whole launch (see launch_firpfbch_col)
    const bool full = nsteps % ((size_t)G * run) == 0;
    const Pow2Plan plan = nranks > 1 ? make_pow2_plan(M / nranks) : Pow2Plan{0, {0}};
    const int rk = nranks > 1 ? rank : 0, nr = nranks > 1 ? nranks : 1;
    const unsigned grid = (unsigned)nblk;
#define YG_C5_LAUNCH(SH, FU) firpfbch2_col_kernel<P, LGM, SH, FU><<<grid, 256, lds, st>>>(fh, hist_len, fx, h, ftw, plan, rk, nr, fy, nsteps, (int)run, fn, p_real)
    if (nranks > 1) { if (full) YG_C5_LAUNCH(true, true); else YG_C5_LAUNCH(true, false); }
    else { if (full) YG_C5_LAUNCH(false, true); else YG_C5_LAUNCH(false, false); }
#undef YG_C5_LAUNCH
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// firpfbch2, column-sliding form for wide banks (M = 512: 2m <= 8, M = 1024: 2m <= 4; built for 2 / 4 / 8, zero taps
// behind shorter branches): every lane owns M/256 windows
// b = t + 256 cc (so the early/late-fed split b >= M/2 is uniform per cc), half tiles of HS = 4096/M steps.
template <int P, int LGM, bool FULL>                // FULL: every workgroup of the launch holds `run` steps (see firpfbch_col_kernel)
__global__ void __launch_bounds__(256)
firpfbch2_wide_kernel(const float2 *__restrict__ hist, int hist_len, const float2 *__restrict__ x,
                      const float *__restrict__ h, const float2 *__restrict__ twM,
                      float2 *__restrict__ y, size_t nsteps, int run /* steps, multiple of 2*HS */,
                      int p_real /* <= P: the other taps are zero */) {
    constexpr int M = 1 << LGM, lgM = LGM, M2 = M / 2, C = M / 256, HS = 4096 / M, TILE = 2 * HS, HP = HS / 2;
    constexpr int R0 = LGM == 9 ? 8 : 16;
    constexpr int nq = HS, lgnq = LGM == 9 ? 3 : 2;
    constexpr int pitch = M + 32 / nq;
    static_assert((TILE / 2) % P == 0, "ring slots must be static");       // pairs per tile = HS
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *va = reinterpret_cast<float2 *>(smem);               // [HS steps][pitch]
    float2 *vb = va + nq * pitch;
    float2 *twl = vb + nq * pitch;                               // M
    const int t = threadIdx.x;
    for (int e = t; e < M; e += 256) twl[e] = twM[e];
    float h0r[C][P], h1[C][P];
    int pos[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int b = t + 256 * cc;
        const int bpar = (b >= M2) ? 1 : 0;                      // uniform per cc (M2 is a multiple of 256)
        pos[cc] = bpar ? (M - 1 - b) : (M2 - 1 - b);
        const int i1 = (b - M2 + M) & (M - 1);
#pragma unroll
        for (int m = 0; m < P; ++m) {
            const int n0 = bpar ? (m + P - 1) % P : m;           // rotated for the late-fed half
            h0r[cc][m] = n0 < p_real ? h[b + n0 * M] : 0.0f;
            h1[cc][m] = m < p_real ? h[i1 + m * M] : 0.0f;
        }
    }
    const float invM = 1.0f / (float)M;
    const long long x_len = (long long)nsteps * M2;
    const long long s_begin = (long long)blockIdx.x * run;       // even
    const long long left = (long long)nsteps - s_begin;
    const int nvalid = (int)(left < run ? left : run);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + s_begin * M2, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y + s_begin * M, 0xffffffffu);
    float2 w[C][P];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int bpar = (2 * cc >= C) ? 1 : 0;
#pragma unroll
        for (int n = 1; n <= P; ++n)                             // pairs -1 .. -P
            w[cc][(P - n) % P] = load_hist(hist, hist_len, x, (s_begin - 2 * n + bpar) * M2 + pos[cc], x_len);
    }
    float2 xa[C][HP], xb[C][HP];
    auto loadp = [&](float2 (&d)[C][HP], int st /* first step of the half tile */) {
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int bpar = (2 * cc >= C) ? 1 : 0;
            const float2 *xg = x + (s_begin + bpar) * M2 + pos[cc];          // + 2kk*M2 = kk*M
#pragma unroll
            for (int kk = 0; kk < HP; ++kk) {
                if constexpr (FULL)
                    d[cc][kk] = buf_ld_aux<kAnaLoad>(rx, 8u * (unsigned)(bpar * M2 + pos[cc]), 8u * ((unsigned)(st / 2 + kk) << lgM));
                else
                    d[cc][kk] = (st + 2 * kk + bpar < nvalid) ? xg[(unsigned)(st / 2 + kk) << lgM] : make_float2(0.f, 0.f);
            }
        }
    };
    float2 *yb = y + s_begin * M;
    auto half_tile = [&](float2 (&xin)[C][HP], int st, auto slot0) {
        constexpr int S0 = decltype(slot0)::value;               // ring slot of the half tile's first pair
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const bool late = 2 * cc >= C;
#pragma unroll
            for (int kk = 0; kk < HP; ++kk) {
                const float2 old = w[cc][(S0 + kk) % P];
                w[cc][(S0 + kk) % P] = late ? old : xin[cc][kk];  // even step: the late-fed half still sees the old sample
                float2 a0 = make_float2(0.f, 0.f), a1 = make_float2(0.f, 0.f);
#pragma unroll
                for (int m = 0; m < P; ++m) {
                    const float2 sv = w[cc][(S0 + kk - m + 4 * P) % P];
                    a0.x = fmaf(sv.x, h0r[cc][m], a0.x);
                    a0.y = fmaf(sv.y, h0r[cc][m], a0.y);
                }
                w[cc][(S0 + kk) % P] = xin[cc][kk];               // odd step: everybody is fed
#pragma unroll
                for (int m = 0; m < P; ++m) {
                    const float2 sv = w[cc][(S0 + kk - m + 4 * P) % P];
                    a1.x = fmaf(sv.x, h1[cc][m], a1.x);
                    a1.y = fmaf(sv.y, h1[cc][m], a1.y);
                }
                va[(2 * kk) * pitch + t + 256 * cc] = a0;
                va[(2 * kk + 1) * pitch + t + 256 * cc] = a1;
            }
        }
        if (st + TILE < run) loadp(xin, st + TILE);
        __syncthreads();
        stockham_pass<R0, +1, true>(va, vb, M, 1, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
        stockham_pass<8, +1, true>(vb, va, M, R0, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
        stockham_pass<8, +1, true>(va, vb, M, R0 * 8, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < HS * C; ++i) {
            const int e = t + 256 * i;
            const int q = e >> lgM, k = e & (M - 1);
            if constexpr (FULL) {
                const float2 v = vb[q * pitch + k];
                buf_st_aux<kAnaStore>(ry, 8u * (((unsigned)q << lgM) + k), 8u * ((unsigned)st << lgM), make_float2(v.x * invM, v.y * invM));
            } else if (st + q < nvalid) {
                const float2 v = vb[q * pitch + k];
                yb[((unsigned)(st + q) << lgM) + k] = make_float2(v.x * invM, v.y * invM);
            }
        }
        __syncthreads();
    };
    loadp(xa, 0);
    loadp(xb, HS);
    for (int s0 = 0; s0 < run; s0 += TILE) {
        half_tile(xa, s0, std::integral_constant<int, 0>{});
        half_tile(xb, s0 + HS, std::integral_constant<int, HP>{});
    }
}

template <int P, int LGM>
static int launch_firpfbch2_wide(const cf32 *hist, int hist_len, const cf32 *x, const float *h, const cf32 *twM,
                                 cf32 *y, size_t nsteps, hipStream_t st, int p_real) {
    constexpr int M = 1 << LGM, HS = 4096 / M, TILE = 2 * HS;
    size_t run = nsteps / kColWgs;
    run = run / TILE * TILE;
    if (run < (size_t)TILE) run = TILE;
    if (run > 512) run = 512;
    const size_t nblk = (nsteps + run - 1) / run;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const size_t lds = (2 * (size_t)HS * (M + 32 / HS) + (size_t)M) * sizeof(float2);
    static bool raised = false;
    if (!raised) {
        YG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(firpfbch2_wide_kernel<P, LGM, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        YG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(firpfbch2_wide_kernel<P, LGM, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y);
    if (nsteps % run == 0)
        firpfbch2_wide_kernel<P, LGM, true><<<(unsigned)nblk, 256, lds, st>>>(fh, hist_len, fx, h, ftw, fy, nsteps, (int)run, p_real);
    else
        firpfbch2_wide_kernel<P, LGM, false><<<(unsigned)nblk, 256, lds, st>>>(fh, hist_len, fx, h, ftw, fy, nsteps, (int)run, p_real);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

int launch_firpfbch2(const cf32 *hist, int hist_len, const cf32 *x, const float *h, int M, int m,
                     const cf32 *twM, uint64_t step0, int rank, int nranks, cf32 *y, size_t nsteps,
                     hipStream_t st, cf32 *hist_next, bool *hist_written) {
    if (hist_written) *hist_written = false;
    if (nsteps == 0) return YAGI_OK;
    const int p = 2 * m, M2 = M / 2;
    if (nranks < 1 || M % nranks || rank < 0 || rank >= nranks)
        return fail(YAGI_ERR_CONFIG, "firpfbch2: %d channels do not shard over %d ranks", M, nranks);
    const size_t lead = (size_t)(p - 1) * M + M2;
    if ((size_t)hist_len != lead) return fail(YAGI_ERR_INTERNAL, "firpfbch2: bad history length");
    if (nranks == 1 && (step0 & 1) == 0 && nsteps >= 64 &&
        ((M == 512 && p <= 8) || (M == 1024 && p <= 4))) {           // p = 6: zero taps behind the real ones
        if (M == 512)
            return p <= 2 ? launch_firpfbch2_wide<2, 9>(hist, hist_len, x, h, twM, y, nsteps, st, p)
                 : p <= 4 ? launch_firpfbch2_wide<4, 9>(hist, hist_len, x, h, twM, y, nsteps, st, p)
                          : launch_firpfbch2_wide<8, 9>(hist, hist_len, x, h, twM, y, nsteps, st, p);
        return p <= 2 ? launch_firpfbch2_wide<2, 10>(hist, hist_len, x, h, twM, y, nsteps, st, p)
                      : launch_firpfbch2_wide<4, 10>(hist, hist_len, x, h, twM, y, nsteps, st, p);
    }
    if ((M == 8 || M == 16 || M == 32 || M == 64 || M == 128 || M == 256) && (step0 & 1) == 0 && nsteps >= 64 &&
        is_pow2(M / nranks)) {
#define YG_COL2_CASE(PP)                                                                                          \
    case PP:                                                                                                      \
        if (hist_written) *hist_written = hist_next != nullptr;                                                   \
        return M == 8 ? launch_firpfbch2_col<PP, 3>(hist, hist_len, x, h, twM, rank, nranks, y, nsteps, st, hist_next, p)       \
             : M == 16 ? launch_firpfbch2_col<PP, 4>(hist, hist_len, x, h, twM, rank, nranks, y, nsteps, st, hist_next, p)      \
             : M == 32 ? launch_firpfbch2_col<PP, 5>(hist, hist_len, x, h, twM, rank, nranks, y, nsteps, st, hist_next, p)      \
             : M == 64 ? launch_firpfbch2_col<PP, 6>(hist, hist_len, x, h, twM, rank, nranks, y, nsteps, st, hist_next, p)      \
             : M == 128 ? launch_firpfbch2_col<PP, 7>(hist, hist_len, x, h, twM, rank, nranks, y, nsteps, st, hist_next, p)     \
                        : launch_firpfbch2_col<PP, 8>(hist, hist_len, x, h, twM, rank, nranks, y, nsteps, st, hist_next, p);
        // branch lengths between the built sizes (2m = 6, 10, 12, 14) take the next one with zero taps behind theirs
        switch (p <= 2 ? 2 : p <= 4 ? 4 : p <= 8 ? 8 : p <= 16 ? 16 : 0) {
            YG_COL2_CASE(2)
            YG_COL2_CASE(4)
            YG_COL2_CASE(8)
            YG_COL2_CASE(16)
            default: break;
        }
#undef YG_COL2_CASE
    }
    int S = 4096 / M;
    if (S < 1) S = 1;
    const size_t fixed = (size_t)M * sizeof(float2) + (size_t)p * M * sizeof(float);
    auto need = [&](int s) { return 2 * (size_t)s * M * sizeof(float2) + fixed; };
    const size_t budget = M >= 512 ? 78 * 1024 : kChanLdsBudget;
    while (S > 1 && need(S) > budget) S /= 2;
    if (need(S) > 150 * 1024) return fail(YAGI_ERR_CONFIG, "firpfbch2: M*m too large for LDS (%d x %d)", M, m);
    const int Mr = M / nranks;
    const bool pow2 = is_pow2(Mr);
    const void *fn = pow2 ? reinterpret_cast<const void *>(firpfbch2_kernel<true>)
                          : reinterpret_cast<const void *>(firpfbch2_kernel<false>);
    if (need(S) > 64 * 1024) YG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t tiles = (nsteps + S - 1) / S;
    const unsigned grid = (unsigned)(tiles < 65536 ? tiles : 65536);
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y);
    if (pow2)
        firpfbch2_kernel<true><<<grid, 256, need(S), st>>>(fh, hist_len, fx, h, M, p, ftw, FacList{0, {0}}, make_pow2_plan(Mr),
                                                          (unsigned long long)step0, rank, nranks, fy, nsteps, S);
    else
        firpfbch2_kernel<false><<<grid, 256, need(S), st>>>(fh, hist_len, fx, h, M, p, ftw, factorize_small(Mr), Pow2Plan{0, {0}},
                                                           (unsigned long long)step0, rank, nranks, fy, nsteps, S);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// firpfbch2 synthesizer (liquid-dsp firpfbch2_crcf_execute_synthesizer semantics; absent from the reference):
// one step = M channel samples -> M/2 output samples.  With v_s = IDFT_M(X_s) / 2 (unnormalised inverse DFT; the 1/2
// gives analyzer -> synthesizer unit gain) and f = parity of the step index, b = i + f M/2:
//     y[s M/2 + i] = sum_{n < 2m} h[i + n M] v_{s-2n}[b]  +  sum_{n < 2m} h[i + M/2 + n M] v_{s-1-2n}[b],   i < M/2
// (the two window sets of liquid's implementation are the even- and odd-step inverse transforms).  A workgroup
// inverse-transforms the S steps of its tile and the 4m - 1 steps before them in LDS, then evaluates the dot products.
// With the analyzer's prototype kaiser(2Mm+1, 1/M) and the synthesizer's kaiser(2Mm+1, 0.5/M), both scaled to sum M,
// synthesizer(analyzer(x)) reproduces x delayed by 2Mm - M/2 + 1 samples (-60 dB at m = 3, -87 dB at m = 4, As = 80).
// ---------------------------------------------------------------------------------------------
template <bool POW2>
__global__ void __launch_bounds__(256)
firpfbch2_syn_kernel(const float2 *__restrict__ hist, int hist_len, const float2 *__restrict__ x,
                     const float *__restrict__ h, int M, int p /* 2m */, const float2 *__restrict__ twM,
                     FacList fl, Pow2Plan plan, unsigned long long step0, float2 *__restrict__ y,
                     size_t nsteps, int S) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int M2 = M / 2, back = 2 * p - 1;                     // steps of history one output reaches
    float2 *va = reinterpret_cast<float2 *>(smem);              // (S + back) * M
    float2 *vb = va + (size_t)(S + back) * M;
    float2 *twl = vb + (size_t)(S + back) * M;                  // M
    for (int e = threadIdx.x; e < M; e += 256) twl[e] = twM[e];
    const size_t s0 = (size_t)blockIdx.x * S;
    const int ns = (int)((nsteps - s0) < (size_t)S ? (nsteps - s0) : (size_t)S);
    const long long base = ((long long)s0 - back) * M;          // first channel sample of the tile's span
    const int ntr = ns + back, nspan = ntr * M;
    const long long x_len = (long long)nsteps * M;
    if (base >= 0 && base + nspan <= x_len) {
        const float2 *src = x + base;
        for (int u = threadIdx.x; u < nspan; u += 256) va[u] = src[u];
    } else {
        for (int u = threadIdx.x; u < nspan; u += 256) va[u] = load_hist(hist, hist_len, x, base + u, x_len);
    }
    __syncthreads();
    const float2 *res = POW2 ? lds_fft_pow2<+1>(va, vb, M, ntr, plan, twl, 1, true)
                             : lds_dft_frames(va, vb, M, ntr, fl, twl, 1, true);
    for (int e = threadIdx.x; e < ns * M2; e += 256) {
        const int sl = e / M2, i = e - sl * M2;
        const int f = (int)((step0 + s0 + sl) & 1ull);
        const int b = i + f * M2;
        const float2 *vp = res + (size_t)(sl + back) * M + b;   // v_s[b]; v_{s-k}[b] is k*M before it
        float2 acc = make_float2(0.f, 0.f);
        for (int n = 0; n < p; ++n) {
            const float h0 = h[i + n * M], h1 = h[i + M2 + n * M];
            const float2 a0 = vp[-(2 * n) * M], a1 = vp[-(2 * n + 1) * M];
            acc.x = fmaf(a0.x, h0, acc.x);
            acc.y = fmaf(a0.y, h0, acc.y);
            acc.x = fmaf(a1.x, h1, acc.x);
            acc.y = fmaf(a1.y, h1, acc.y);
        }
        y[s0 * M2 + e] = make_float2(0.5f * acc.x, 0.5f * acc.y);
    }
}

// Column-sliding firpfbch2 synthesizer (M in {8..256}, m <= 4: rings of 8 and 16 steps, 4m zero-padded to them):
// firpfbch_syn_col_kernel with a ring of 4m steps per
// column.  Column c of the inverse-transformed steps feeds output i = c mod M/2 -- on the steps whose parity f selects
// its half (b = i + f M/2 = c) -- with the taps of lag k: hk[k] = h[i + (k & 1) M/2 + (k >> 1) M] (lag 2n: first sum, lag
// 2n + 1: second sum; added in lag order like the tiled kernel).  For M >= 128 a wave is all-lower or all-upper half:
// the two halves of a workgroup alternate, nobody diverges.  A run starts with W = 8 (m = 2) or 16 warm-up steps.
template <int P2, int LGM, bool FAST>
__global__ void __launch_bounds__(256)
firpfbch2_syn_col_kernel(const float2 *__restrict__ hist, const float2 *__restrict__ x,
                         const float *__restrict__ h, const float2 *__restrict__ twM,
                         unsigned long long step0, float2 *__restrict__ y, size_t nsteps, int run,
                         int p2_real /* 4m <= P2: the other lags are zero */) {
    constexpr int M = 1 << LGM, lgM = LGM, M2 = M / 2;
    constexpr int R0 = (LGM == 3 || LGM == 5 || LGM == 6) ? 8 : 16, R1 = M / R0;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int G = 256 / M;
    constexpr int nq = G * kColHalf, lgnq = 8 - LGM + 3;
    constexpr int pitch = col_pitch(M, nq);
    constexpr int W = P2 <= 8 ? 8 : 16;                          // warm-up steps per run (>= P2 - 1)
    float2 *va = reinterpret_cast<float2 *>(smem);
    float2 *vb = va + nq * pitch;
    float2 *twl = vb + nq * pitch;
    const int g = threadIdx.x >> lgM, c = threadIdx.x & (M - 1);
    const int i = c & (M2 - 1), hi = c >> (LGM - 1);
    for (int e = threadIdx.x; e < M; e += 256) twl[e] = twM[e];
    float hk[P2];
#pragma unroll
    for (int k = 0; k < P2; ++k) hk[k] = k < p2_real ? 0.5f * h[i + (k & 1) * M2 + (k >> 1) * M] : 0.0f;
    const int hist_len = (p2_real - 1) * M;
    const long long x_len = (long long)nsteps * M;
    const long long wg_first = (long long)blockIdx.x * G * run;
    const long long s_begin = wg_first + (long long)g * run;
    const long long left = (long long)nsteps - s_begin;
    const int nvalid = (int)(left < 0 ? 0 : (left < run ? left : run));
    const int par = (int)((step0 + (unsigned long long)s_begin) & 1ull) ^ hi;   // this lane's outputs: steps t with (t + par) even
    // FAST (as in firpfbch_syn_col_kernel): buffer descriptors, no history or range checks
    const bool first_wg = blockIdx.x == 0;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(FAST && !first_wg ? x + (wg_first - W) * M : x, 0xffffffffu);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(FAST ? y + wg_first * M2 : y, 0xffffffffu);
    const unsigned vx = 8u * ((unsigned)(g * run) * M + c), vy = 8u * ((unsigned)(g * run) * M2 + i);
    const unsigned sx0 = first_wg ? 0u : 8u * ((unsigned)W << lgM);
    float2 w[P2];
#pragma unroll
    for (int n = 0; n < P2; ++n) w[n] = make_float2(0.f, 0.f);
    auto half_tile = [&](int t, auto slot0) {
        constexpr int S0 = decltype(slot0)::value;
#pragma unroll
        for (int j = 0; j < kColHalf; ++j) {
            if (FAST && (t >= 0 || !first_wg)) {
                va[(g * kColHalf + j) * pitch + c] = buf_ld(rx, vx, sx0 + 8u * (unsigned)((t + j) << lgM));
            } else {
                const long long s = s_begin + t + j;
                va[(g * kColHalf + j) * pitch + c] = (t + j < nvalid) ? load_hist(hist, hist_len, x, s * M + c, x_len)
                                                                     : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        stockham_pass<R0, +1, true>(va, vb, M, 1, nq, twl, 1, true, pitch, lgnq);
        __syncthreads();
        const float2 *res = vb;
        if constexpr (R1 > 1) {
            stockham_pass<R1, +1, true>(vb, va, M, R0, nq, twl, 1, true, pitch, lgnq);
            __syncthreads();
            res = va;
        }
#pragma unroll
        for (int j = 0; j < kColHalf; ++j) {
            w[(S0 + j) % P2] = res[(g * kColHalf + j) * pitch + c];
            if (((j + par) & 1) == 0 && t + j >= 0 && (FAST || t + j < nvalid)) {   // t is even: parity of the step = parity of j
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int k = 0; k < P2; ++k) {
                    const float2 sv = w[(S0 + j - k + 4 * P2) % P2];
                    acc.x = fmaf(sv.x, hk[k], acc.x);
                    acc.y = fmaf(sv.y, hk[k], acc.y);
                }
                if constexpr (FAST) buf_st(ry, vy, 8u * ((unsigned)(t + j) << (lgM - 1)), acc);
                else y[(s_begin + t + j) * M2 + i] = acc;
            }
        }
        __syncthreads();                             // the next half tile overwrites va / vb
    };
    if constexpr (P2 <= 8) {
        for (int t0 = -W; t0 < run; t0 += kColHalf) half_tile(t0, std::integral_constant<int, 0>{});
    } else {
        for (int t0 = -W; t0 < run; t0 += kColTile) {
            half_tile(t0, std::integral_constant<int, 0>{});
            half_tile(t0 + kColHalf, std::integral_constant<int, kColHalf>{});
        }
    }
}

template <int P2, int LGM>
static int launch_firpfbch2_syn_col(const cf32 *hist, const cf32 *x, const float *h, const cf32 *twM, uint64_t step0,
                                    cf32 *y, size_t nsteps, hipStream_t st, int p2_real) {
    constexpr int M = 1 << LGM;
    const int G = 256 / M;
    size_t run = nsteps / (kColWgs * G);
    run = run / kColTile * kColTile;
    if (run < (size_t)kColTile) run = kColTile;
    if (run > 256) run = 256;
    const size_t ngroups = (nsteps + run - 1) / run;
    const size_t nblk = (ngroups + G - 1) / G;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const size_t lds = (2 * (size_t)G * kColHalf * col_pitch(M, G * kColHalf) + (size_t)M) * sizeof(float2);
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y);
    const unsigned long long s0 = (unsigned long long)step0;
    if (nsteps % ((size_t)G * run) == 0)             // every workgroup full (see launch_firpfbch_col)
        firpfbch2_syn_col_kernel<P2, LGM, true><<<(unsigned)nblk, 256, lds, st>>>(fh, fx, h, ftw, s0, fy, nsteps, (int)run, p2_real);
    else
        firpfbch2_syn_col_kernel<P2, LGM, false><<<(unsigned)nblk, 256, lds, st>>>(fh, fx, h, ftw, s0, fy, nsteps, (int)run, p2_real);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

int launch_firpfbch2_syn(const cf32 *hist, int hist_len, const cf32 *x, const float *h, int M, int m,
                         const cf32 *twM, uint64_t step0, cf32 *y, size_t nsteps, hipStream_t st) {
    if (nsteps == 0) return YAGI_OK;
    const int p = 2 * m, back = 2 * p - 1;
    if (hist_len != back * M) return fail(YAGI_ERR_INTERNAL, "firpfbch2 synthesizer: bad history length");
    if ((M == 8 || M == 16 || M == 32 || M == 64 || M == 128 || M == 256) && m <= 4 && nsteps >= 64) {
#define YG_SYN2_CASE(PP)                                                                                     \
    (M == 8 ? launch_firpfbch2_syn_col<PP, 3>(hist, x, h, twM, step0, y, nsteps, st, 2 * p)                         \
     : M == 16 ? launch_firpfbch2_syn_col<PP, 4>(hist, x, h, twM, step0, y, nsteps, st, 2 * p)                      \
     : M == 32 ? launch_firpfbch2_syn_col<PP, 5>(hist, x, h, twM, step0, y, nsteps, st, 2 * p)                      \
     : M == 64 ? launch_firpfbch2_syn_col<PP, 6>(hist, x, h, twM, step0, y, nsteps, st, 2 * p)                      \
     : M == 128 ? launch_firpfbch2_syn_col<PP, 7>(hist, x, h, twM, step0, y, nsteps, st, 2 * p)                     \
                : launch_firpfbch2_syn_col<PP, 8>(hist, x, h, twM, step0, y, nsteps, st, 2 * p))
        return m <= 2 ? YG_SYN2_CASE(8) : YG_SYN2_CASE(16);     // 4m lags zero-padded to the ring of 8 or 16
#undef YG_SYN2_CASE
    }
    int S = 4096 / M;
    if (S < 1) S = 1;
    auto need = [&](int s) { return (2 * (size_t)(s + back) * M + (size_t)M) * sizeof(float2); };
    while (S > 1 && need(S) > 78 * 1024) S /= 2;
    if (need(S) > 150 * 1024) return fail(YAGI_ERR_CONFIG, "firpfbch2 synthesizer: M*m too large for LDS (%d x %d)", M, m);
    const bool pow2 = is_pow2(M);
    const void *fn = pow2 ? reinterpret_cast<const void *>(firpfbch2_syn_kernel<true>)
                          : reinterpret_cast<const void *>(firpfbch2_syn_kernel<false>);
    if (need(S) > 64 * 1024) YG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t tiles = (nsteps + S - 1) / S;
    if (tiles > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const float2 *fh = reinterpret_cast<const float2 *>(hist), *fx = reinterpret_cast<const float2 *>(x);
    const float2 *ftw = reinterpret_cast<const float2 *>(twM);
    float2 *fy = reinterpret_cast<float2 *>(y);
    if (pow2)
        firpfbch2_syn_kernel<true><<<(unsigned)tiles, 256, need(S), st>>>(fh, hist_len, fx, h, M, p, ftw, FacList{0, {0}}, make_pow2_plan(M),
                                                                         (unsigned long long)step0, fy, nsteps, S);
    else
        firpfbch2_syn_kernel<false><<<(unsigned)tiles, 256, need(S), st>>>(fh, hist_len, fx, h, M, p, ftw, factorize_small(M), Pow2Plan{0, {0}},
                                                                          (unsigned long long)step0, fy, nsteps, S);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// gathered[rank][step][q]  ->  y[step][rank + nranks*q]
__global__ void __launch_bounds__(256)
assemble_kernel(const float2 *__restrict__ g, size_t nsteps, int M, int R, float2 *__restrict__ y) {
    const int Mr = M / R;
    const size_t total = nsteps * (size_t)M;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const size_t s = e / M;
        const int k = (int)(e - s * M);
        const int r = k % R, q = k / R;
        y[e] = g[((size_t)r * nsteps + s) * Mr + q];
    }
}

int launch_firpfbch2_assemble(const cf32 *gathered, size_t nsteps, int M, int nranks, cf32 *y,
                              hipStream_t st) {
    const size_t total = nsteps * (size_t)M;
    if (total == 0) return YAGI_OK;
    if (nranks < 1 || M % nranks) return fail(YAGI_ERR_CONFIG, "assemble: bad nranks");
    size_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    assemble_kernel<<<(unsigned)g, 256, 0, st>>>(reinterpret_cast<const float2 *>(gathered), nsteps, M,
                                                nranks, reinterpret_cast<float2 *>(y));
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

}  // namespace yagi

