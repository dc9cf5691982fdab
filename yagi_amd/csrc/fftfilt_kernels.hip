// fftfilt_kernels.hip -- the element-wise stages of FftFilt<T,Coeff>::execute
// (src/filter/fftfilt.rs:103-138): zero-padded copy into the 2n-point time buffer, the product with
// FFT{h}, and the overlap-add with the previous block's tail.  The two 2n-point transforms between
// them are launch_fft_batch (fft_kernels.hip), so B consecutive blocks run as one batch:
//   time[b] = [x_b ; 0]  ->  F_b = FFT(time[b]) * H  ->  t_b = IFFT(F_b)
//   y_b[i]  = from_complex((t_b[i] + t_{b-1}[n+i]) * scale),   t_{-1}[n..2n) = w (state)
//   w'      = t_{B-1}[n..2n)
#include "devmath.hpp"
#include "kernels.hpp"

namespace yagi {

__device__ __forceinline__ float2 to_cx(float v) { return make_float2(v, 0.f); }
__device__ __forceinline__ float2 to_cx(cf32 v) { return make_float2(v.re, v.im); }
__device__ __forceinline__ void from_cx(float2 c, float *o) { *o = c.x; }       // FromComplex32 for f32: re()
__device__ __forceinline__ void from_cx(float2 c, cf32 *o) { *o = cf32{c.x, c.y}; }

template <class T>
__global__ void __launch_bounds__(256)
fftfilt_pad_kernel(const T *__restrict__ x, int n, size_t nblocks, float2 *__restrict__ time) {
    const size_t total = nblocks * 2 * (size_t)n;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / (2 * (size_t)n);
        const int i = (int)(e - b * 2 * (size_t)n);
        time[e] = (i < n) ? to_cx(x[b * n + i]) : make_float2(0.f, 0.f);
    }
}

__global__ void __launch_bounds__(256)
fftfilt_mul_kernel(float2 *__restrict__ freq, const float2 *__restrict__ hf, int n2, size_t nblocks) {
    const size_t total = nblocks * (size_t)n2;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
        freq[e] = cmul(freq[e], hf[e % n2]);
}

template <class T, class C>
__global__ void __launch_bounds__(256)
fftfilt_ola_kernel(const float2 *__restrict__ t, const float2 *__restrict__ w, int n, size_t nblocks,
                   C scale, T *__restrict__ y, float2 *__restrict__ w_next) {
    const size_t total = nblocks * (size_t)n;
    const float2 sc = to_cx(scale);
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / n;
        const int i = (int)(e - b * n);
        const float2 cur = t[b * 2 * (size_t)n + i];
        const float2 prev = (b == 0) ? w[i] : t[(b - 1) * 2 * (size_t)n + n + i];
        from_cx(cmul(cadd(cur, prev), sc), &y[e]);
        if (b == nblocks - 1) w_next[i] = t[b * 2 * (size_t)n + n + i];
    }
}

static unsigned ew_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <class T>
int launch_fftfilt_pad(const T *x, int n, size_t nblocks, cf32 *time, hipStream_t st) {
    fftfilt_pad_kernel<T><<<ew_grid(nblocks * 2 * (size_t)n), 256, 0, st>>>(x, n, nblocks, reinterpret_cast<float2 *>(time));
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
int launch_fftfilt_mul(cf32 *freq, const cf32 *hf, int n2, size_t nblocks, hipStream_t st) {
    fftfilt_mul_kernel<<<ew_grid(nblocks * (size_t)n2), 256, 0, st>>>(reinterpret_cast<float2 *>(freq),
                                                                     reinterpret_cast<const float2 *>(hf), n2, nblocks);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
template <class T, class C>
int launch_fftfilt_ola(const cf32 *t, const cf32 *w, int n, size_t nblocks, C scale, T *y, cf32 *w_next,
                       hipStream_t st) {
    fftfilt_ola_kernel<T, C><<<ew_grid(nblocks * (size_t)n), 256, 0, st>>>(
        reinterpret_cast<const float2 *>(t), reinterpret_cast<const float2 *>(w), n, nblocks, scale, y,
        reinterpret_cast<float2 *>(w_next));
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

template int launch_fftfilt_pad<float>(const float *, int, size_t, cf32 *, hipStream_t);
template int launch_fftfilt_pad<cf32>(const cf32 *, int, size_t, cf32 *, hipStream_t);
template int launch_fftfilt_ola<float, float>(const cf32 *, const cf32 *, int, size_t, float, float *, cf32 *, hipStream_t);
template int launch_fftfilt_ola<cf32, float>(const cf32 *, const cf32 *, int, size_t, float, cf32 *, cf32 *, hipStream_t);
template int launch_fftfilt_ola<cf32, cf32>(const cf32 *, const cf32 *, int, size_t, cf32, cf32 *, cf32 *, hipStream_t);

}  // namespace yagi
