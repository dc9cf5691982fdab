// spgram_kernels.hip -- the device stages of fft::Spgram (src/fft/spgram.rs:237-316):
//   step():       buf_time[k] = buffer.read()[k] * w[k] (k < window_len), zero padded to nfft  :263-268
//                 -> Fft::run (launch_fft_batch)                                               :271
//                 psd[i] = first ? |X_i|^2 : gamma*psd[i] + alpha*|X_i|^2                      :277-284
//   get_psd_mag / get_psd: fft-shifted, max(psd, 1e-12) * scale [, 10 log10]                   :292-316
// A write() of n samples triggers a transform every `delay` samples; all transforms of one call are
// built, transformed and accumulated as one batch (frames in stream order, so the accumulation order
// per bin is the reference's).
#include <cmath>

#include "fft_radix.hpp"
#include "kernels.hpp"

namespace yagi {

__device__ __forceinline__ float2 sp_cx(float v, float w) { return make_float2(v * w, 0.f); }
__device__ __forceinline__ float2 sp_cx(cf32 v, float w) { return make_float2(v.re * w, v.im * w); }

// frame f ends at X index first + f*delay (X = window ++ x, X[0] = x[0]); window holds `wlen` samples
template <class T>
__global__ void __launch_bounds__(256)
spgram_frames_kernel(const T *__restrict__ win, const T *__restrict__ x, const float *__restrict__ w,
                     int wlen, int nfft, long long first, int delay, size_t nframes,
                     float2 *__restrict__ time) {
    const size_t total = nframes * (size_t)nfft;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t f = e / nfft;
        const int k = (int)(e - f * nfft);
        float2 v = make_float2(0.f, 0.f);
        if (k < wlen) {
            const long long idx = first + (long long)f * delay - (wlen - 1) + k;
            const T s = (idx < 0) ? win[wlen + idx] : x[idx];
            v = sp_cx(s, w[k]);
        }
        time[e] = v;
    }
}

// psd[i] <- the recurrence p = gamma p + alpha |X_f[i]|^2 over the nframes of the batch, in stream order
// (first frame ever: p = |X_0[i]|^2).  The recurrence is linear, so it is evaluated in two stages with a fixed
// (deterministic) association: Horner over slabs of kSpSlab consecutive frames in parallel, then Horner over
// the slab results:  p <- gamma^len(s) p + q_s.  (One thread per bin walking all frames serially was 95 % of a
// write() call.)
constexpr int kSpSlab = 32;

__global__ void __launch_bounds__(256)
spgram_accum_slab_kernel(const float2 *__restrict__ freq, int nfft, size_t nframes, float alpha, float gamma,
                         int first_ever, float *__restrict__ part) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t s = blockIdx.y;
    if (i >= nfft) return;
    const size_t f0 = s * kSpSlab;
    const size_t f1 = (f0 + kSpSlab < nframes) ? f0 + kSpSlab : nframes;
    float q = 0.0f;
    for (size_t f = f0; f < f1; ++f) {
        const float2 c = freq[f * nfft + i];
        const float mag = c.x * c.x + c.y * c.y;          // (X * conj X).re
        q = (first_ever && f == 0) ? mag : gamma * q + alpha * mag;
    }
    part[s * nfft + i] = q;
}

__global__ void __launch_bounds__(256)
spgram_accum_final_kernel(const float *__restrict__ part, int nfft, size_t nslabs, size_t nframes, float gpow_full,
                          float gpow_last, int first_ever, float *__restrict__ psd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfft) return;
    float p = psd[i];
    for (size_t s = 0; s < nslabs; ++s) {
        const float g = (s + 1 == nslabs) ? gpow_last : gpow_full;       // gamma^(frames in slab s)
        const float q = part[s * nfft + i];
        p = (first_ever && s == 0) ? q : g * p + q;
    }
    psd[i] = p;
}

__global__ void __launch_bounds__(256)
spgram_psd_kernel(const float *__restrict__ psd, int nfft, float scale, int in_db, float *__restrict__ out) {
    const int nfft_2 = nfft / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nfft; i += gridDim.x * blockDim.x) {
        const int k = (i + nfft_2) % nfft;
        const float v = fmaxf(psd[k], 1e-12f) * scale;          // SPGRAM_PSD_MIN
        out[i] = in_db ? 10.0f * log10f(v) : v;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused form for nfft = 4096 (spgram.rs:263-284 in one kernel): a workgroup walks `slab` consecutive
// transforms; each is tapered while it is loaded (lane t holds window taps 256a + t in registers), transformed
// in registers (fft4096_passes_to_regs) and its |X|^2 added into 16 per-lane accumulators with the weight the
// recurrence p = gamma p + alpha |X|^2 gives frame f of the batch: alpha gamma^(nframes-1-f) (the first frame
// ever: gamma^(nframes-1)).  Neither the tapered frames nor the spectra touch HBM: the input is read once
// (overlapping frames hit in L2) and one float per bin and slab is written.  spgram_sum_kernel then adds the
// slab partials in a fixed order and applies gamma^nframes to the previous estimate.
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256)
spgram_fused4096_kernel(const T *__restrict__ win, const T *__restrict__ x, const float *__restrict__ w, int wlen,
                        long long first, long long x_len, int delay, unsigned nframes, unsigned slab, float alpha,
                        float log2_gamma, int first_ever, const float2 *__restrict__ tw, float *__restrict__ part) {
    __shared__ float2 lds[kFft4096LdsFloat2];
    const unsigned t = threadIdx.x;
    float wt[16];
#pragma unroll
    for (unsigned a = 0; a < 16; ++a) wt[a] = (int)(256u * a + t) < wlen ? w[256u * a + t] : 0.0f;
    float acc[16];
#pragma unroll
    for (unsigned d = 0; d < 16; ++d) acc[d] = 0.0f;
    const unsigned f0 = blockIdx.x * slab;
    const unsigned f1 = (f0 + slab < nframes) ? f0 + slab : nframes;
#pragma unroll 1
    for (unsigned f = f0; f < f1; ++f) {
        // offset the optimiser cannot see through: otherwise the twiddle loads AND their products are hoisted
        // out of the frame loop and sit in ~60 VGPRs
        unsigned z = 0;
        asm volatile("" : "+s"(z));
        const long long base = first + (long long)f * delay - (wlen - 1);       // stream index of frame sample 0
        float2 v[16];
        if (base >= 0 && base + 4096 <= x_len) {           // block-uniform: all 4096 slots of the frame lie in x
            const T *src = x + base;                       // uniform base + 32-bit lane offset, unconditional loads
#pragma unroll
            for (unsigned a = 0; a < 16; ++a) {
                const unsigned k = 256u * a + t;
                const float2 sv = sp_cx(src[k], wt[a]);
                v[a] = (int)k < wlen ? sv : make_float2(0.f, 0.f);     // a select, not a branch
            }
        } else {                                           // reaches into the window: rolled loop through LDS
#pragma unroll 1
            for (int k = (int)t; k < 4096; k += 256) {
                float2 sv = make_float2(0.f, 0.f);
                if (k < wlen) {
                    const long long idx = base + k;
                    sv = sp_cx((idx < 0) ? win[wlen + idx] : x[idx], w[k]);
                }
                lds[k] = sv;
            }
            __syncthreads();
#pragma unroll
            for (unsigned a = 0; a < 16; ++a) v[a] = lds[256u * a + t];
            __syncthreads();
        }
        fft4096_passes_to_regs<-1>(v, lds, tw + z);
        float wgt = exp2f(log2_gamma * (float)(nframes - 1 - f));
        if (!(first_ever && f == 0)) wgt *= alpha;
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) acc[d] = fmaf(wgt, v[d].x * v[d].x + v[d].y * v[d].y, acc[d]);
    }
    float *dst = part + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (unsigned d = 0; d < 16; ++d) dst[t + 256u * d] = acc[d];
}

// The same for nfft = 256 M, M in {1, 2, 4, 8} (256 .. 2048): the workgroup transforms B = 16/M consecutive frames
// at a time (fft_n256m_passes_to_regs: lane (tr, u) holds bins u + 16M i + 256 d of frame tr) and keeps one set of
// accumulators per frame slot tr, so it writes B partial rows.
template <class T, int M>
__global__ void __launch_bounds__(256)
spgram_fused_n256m_kernel(const T *__restrict__ win, const T *__restrict__ x, const float *__restrict__ w, int wlen,
                          long long first, long long x_len, int delay, unsigned nframes, unsigned slab, float alpha,
                          float log2_gamma, int first_ever, const float2 *__restrict__ tw, float *__restrict__ part) {
    constexpr int N = 256 * M, LT = 16 * M, B = 16 / M;
    __shared__ float2 lds[kFft4096LdsFloat2];
    const unsigned t = threadIdx.x, tr = t / LT, u = t % LT;
    float wt[16];
#pragma unroll
    for (unsigned a = 0; a < 16; ++a) wt[a] = (int)(LT * a + u) < wlen ? w[LT * a + u] : 0.0f;
    float acc[16];
#pragma unroll
    for (unsigned d = 0; d < 16; ++d) acc[d] = 0.0f;
    const unsigned f0 = blockIdx.x * slab;                       // slab is a multiple of B
    const unsigned f1 = (f0 + slab < nframes) ? f0 + slab : nframes;
#pragma unroll 1
    for (unsigned fi = f0; fi < f1; fi += B) {
        unsigned z = 0;
        asm volatile("" : "+s"(z));                              // see spgram_fused4096_kernel
        const unsigned f = fi + tr;
        const long long base0 = first + (long long)fi * delay - (wlen - 1);     // frame slot 0, sample 0
        float2 v[16];
        if (base0 >= 0 && base0 + (long long)(B - 1) * delay + N <= x_len && fi + B <= f1) {   // block-uniform
            const T *src = x + base0 + (long long)tr * delay;
#pragma unroll
            for (unsigned a = 0; a < 16; ++a) {
                const unsigned k = LT * a + u;
                const float2 sv = sp_cx(src[k], wt[a]);
                v[a] = (int)k < wlen ? sv : make_float2(0.f, 0.f);
            }
        } else {                                                 // window-reaching / last frames: rolled, through LDS
#pragma unroll 1
            for (int e = (int)t; e < 4096; e += 256) {
                const int trr = e / N, k = e - trr * N;
                float2 sv = make_float2(0.f, 0.f);
                if (fi + trr < f1 && k < wlen) {
                    const long long idx = base0 + (long long)trr * delay + k;
                    sv = sp_cx((idx < 0) ? win[wlen + idx] : x[idx], w[k]);
                }
                lds[e] = sv;
            }
            __syncthreads();
#pragma unroll
            for (unsigned a = 0; a < 16; ++a) v[a] = lds[tr * N + LT * a + u];
            __syncthreads();
        }
        fft_n256m_passes_to_regs<-1, M>(v, lds, tw + z);
        __syncthreads();                                         // lds is reused by the next iteration
        float wgt = 0.0f;                                        // frame slots past the end contribute nothing
        if (f < f1) {
            wgt = exp2f(log2_gamma * (float)(nframes - 1 - f));
            if (!(first_ever && f == 0)) wgt *= alpha;
        }
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) acc[d] = fmaf(wgt, v[d].x * v[d].x + v[d].y * v[d].y, acc[d]);
    }
    float *dst = part + ((size_t)blockIdx.x * B + tr) * N + u;
#pragma unroll
    for (int i = 0; i < B; ++i)
#pragma unroll
        for (int d = 0; d < M; ++d) dst[LT * i + 256 * d] = acc[i * M + d];
}

// rows of partial sums -> rows / kSpFold rows (fixed order): out[c][i] = sum_{s in chunk c} part[s][i]
constexpr unsigned kSpFold = 32;
__global__ void __launch_bounds__(256)
spgram_fold_kernel(const float *__restrict__ part, int nfft, unsigned nrows, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned c = blockIdx.y;
    if (i >= nfft) return;
    const unsigned s0 = c * kSpFold, s1 = (s0 + kSpFold < nrows) ? s0 + kSpFold : nrows;
    float q = 0.0f;
    for (unsigned s = s0; s < s1; ++s) q += part[(size_t)s * nfft + i];
    out[(size_t)c * nfft + i] = q;
}

// psd[i] <- (first_ever ? 0 : gpow psd[i]) + sum_s part[s][i]; 4 partial sums per bin, combined in a fixed order
__global__ void __launch_bounds__(256)
spgram_sum_kernel(const float *__restrict__ part, int nfft, unsigned nslabs, float gpow, int first_ever,
                  float *__restrict__ psd) {
    __shared__ float red[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const unsigned ty = threadIdx.x >> 6;
    float q = 0.0f;
    if (i < nfft)
        for (unsigned s = ty; s < nslabs; s += 4) q += part[(size_t)s * nfft + i];
    red[ty][threadIdx.x & 63] = q;
    __syncthreads();
    if (ty == 0 && i < nfft) {
        const float sum = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        psd[i] = first_ever ? sum : gpow * psd[i] + sum;
    }
}

static unsigned sp_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <class T>
int launch_spgram_frames(const T *win, const T *x, const float *w, int wlen, int nfft, long long first,
                         int delay, size_t nframes, cf32 *time, hipStream_t st) {
    spgram_frames_kernel<T><<<sp_grid(nframes * (size_t)nfft), 256, 0, st>>>(
        win, x, w, wlen, nfft, first, delay, nframes, reinterpret_cast<float2 *>(time));
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
template int launch_spgram_frames<float>(const float *, const float *, const float *, int, int, long long, int, size_t, cf32 *, hipStream_t);
template int launch_spgram_frames<cf32>(const cf32 *, const cf32 *, const float *, int, int, long long, int, size_t, cf32 *, hipStream_t);

size_t spgram_accum_scratch_floats(int nfft, size_t nframes) {
    return ((nframes + kSpSlab - 1) / kSpSlab) * (size_t)nfft;
}
int launch_spgram_accum(const cf32 *freq, int nfft, size_t nframes, float alpha, float gamma, bool first_ever,
                        float *psd, float *part, hipStream_t st) {
    if (nframes == 0) return YAGI_OK;
    const size_t nslabs = (nframes + kSpSlab - 1) / kSpSlab;
    const dim3 g1((unsigned)((nfft + 255) / 256), (unsigned)nslabs);
    spgram_accum_slab_kernel<<<g1, 256, 0, st>>>(reinterpret_cast<const float2 *>(freq), nfft, nframes, alpha, gamma,
                                                first_ever ? 1 : 0, part);
    YG_LAUNCH_CHECK();
    const size_t last = nframes - (nslabs - 1) * kSpSlab;
    spgram_accum_final_kernel<<<(unsigned)((nfft + 255) / 256), 256, 0, st>>>(
        part, nfft, nslabs, nframes, std::pow(gamma, (float)kSpSlab), std::pow(gamma, (float)last), first_ever ? 1 : 0, psd);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
// fused paths: nfft = 4096 and nfft = 256 M (M in {1,2,4,8}); part: spgram_fused_scratch_floats(nfft, nframes) floats
bool spgram_fused_supported(int nfft) { return nfft == 256 || nfft == 512 || nfft == 1024 || nfft == 2048 || nfft == 4096; }
static unsigned spgram_fused_slab(int nfft, size_t nframes) {
    const unsigned B = (unsigned)(4096 / nfft);                  // frames transformed together
    size_t sl = nframes / 1024;
    sl = sl < 4 ? 4 : (sl > 32 ? 32 : sl);
    return (unsigned)((sl + B - 1) / B * B);
}
size_t spgram_fused_scratch_floats(int nfft, size_t nframes) {
    const unsigned slab = spgram_fused_slab(nfft, nframes);
    const size_t rows_floats = ((nframes + slab - 1) / slab) * (size_t)4096;      // B rows of nfft floats per workgroup
    return rows_floats + rows_floats / kSpFold + 4096;           // + the folded rows of the two-level sum
}
template <class T>
int launch_spgram_fused(int nfft, const T *win, const T *x, size_t x_len, const float *w, int wlen, long long first,
                        int delay, size_t nframes, float alpha, float gamma, bool first_ever, const cf32 *twn,
                        float *psd, float *part, hipStream_t st) {
    if (nframes == 0) return YAGI_OK;
    if (nframes > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "too many transforms in one call");
    const unsigned slab = spgram_fused_slab(nfft, nframes);
    const unsigned nwg = (unsigned)((nframes + slab - 1) / slab);
    const float l2g = (float)std::log2((double)gamma);
    const float2 *tw = reinterpret_cast<const float2 *>(twn);
#define YG_SP_ARGS win, x, w, wlen, first, (long long)x_len, delay, (unsigned)nframes, slab, alpha, l2g, first_ever ? 1 : 0, tw, part
    switch (nfft) {
        case 4096: spgram_fused4096_kernel<T><<<nwg, 256, 0, st>>>(YG_SP_ARGS); break;
        case 2048: spgram_fused_n256m_kernel<T, 8><<<nwg, 256, 0, st>>>(YG_SP_ARGS); break;
        case 1024: spgram_fused_n256m_kernel<T, 4><<<nwg, 256, 0, st>>>(YG_SP_ARGS); break;
        case 512: spgram_fused_n256m_kernel<T, 2><<<nwg, 256, 0, st>>>(YG_SP_ARGS); break;
        case 256: spgram_fused_n256m_kernel<T, 1><<<nwg, 256, 0, st>>>(YG_SP_ARGS); break;
        default: return fail(YAGI_ERR_INTERNAL, "spgram: no fused kernel for nfft %d", nfft);
    }
#undef YG_SP_ARGS
    YG_LAUNCH_CHECK();
    unsigned nrows = nwg * (unsigned)(4096 / nfft);
    const float *rows = part;
    if (nrows > 2 * kSpFold) {                                   // many rows: fold them 32 to 1 across the whole chip first
        float *folded = part + (size_t)nrows * nfft;
        const unsigned nf = (nrows + kSpFold - 1) / kSpFold;
        spgram_fold_kernel<<<dim3((unsigned)((nfft + 255) / 256), nf), 256, 0, st>>>(part, nfft, nrows, folded);
        YG_LAUNCH_CHECK();
        rows = folded;
        nrows = nf;
    }
    spgram_sum_kernel<<<(unsigned)(nfft / 64), 256, 0, st>>>(rows, nfft, nrows, (float)std::pow((double)gamma, (double)nframes),
                                                            first_ever ? 1 : 0, psd);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
template int launch_spgram_fused<float>(int, const float *, const float *, size_t, const float *, int, long long, int, size_t, float, float, bool, const cf32 *, float *, float *, hipStream_t);
template int launch_spgram_fused<cf32>(int, const cf32 *, const cf32 *, size_t, const float *, int, long long, int, size_t, float, float, bool, const cf32 *, float *, float *, hipStream_t);

int launch_spgram_psd(const float *psd, int nfft, float scale, bool in_db, float *out, hipStream_t st) {
    spgram_psd_kernel<<<sp_grid((size_t)nfft), 256, 0, st>>>(psd, nfft, scale, in_db ? 1 : 0, out);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

}  // namespace yagi
