// spgram_kernels.hip -- the device stages of fft::Spgram (src/fft/spgram.rs:237-316):
//   step():       buf_time[k] = buffer.read()[k] * w[k] (k < window_len), zero padded to nfft  :263-268
//                 -> Fft::run (launch_fft_batch)                                               :271
//                 psd[i] = first ? |X_i|^2 : gamma*psd[i] + alpha*|X_i|^2                      :277-284
//   get_psd_mag / get_psd: fft-shifted, max(psd, 1e-12) * scale [, 10 log10]                   :292-316
// A write() of n samples triggers a transform every `delay` samples; all transforms of one call are
// built, transformed and accumulated as one batch (frames in stream order, so the accumulation order
// per bin is the reference's).
#include <cmath>

#include "devmath.hpp"
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
int launch_spgram_psd(const float *psd, int nfft, float scale, bool in_db, float *out, hipStream_t st) {
    spgram_psd_kernel<<<sp_grid((size_t)nfft), 256, 0, st>>>(psd, nfft, scale, in_db ? 1 : 0, out);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

}  // namespace yagi
