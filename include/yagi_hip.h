/*
 * yagi_hip.h -- C ABI of libyagi_hip.so: the MI355X (gfx950) FIR/FFT engine that sits behind
 * yagi's `new / push / execute / execute_block` object API.
 *
 * yagi (EEGKit/yagi, Rust) has no FFI of its own for this path; its only C-ABI precedent is the
 * stub `c_shim` (c_shim/src/lib.rs:3-12: opaque `*mut foo_s` handle, `c_uint` sizes, `c_int`
 * status).  The entry points below are what a `hip` feature of yagi would bind with
 * `extern "C"` (see INTEGRATION.md for the Rust side).  Every group cites the reference item
 * (file:line, relative to the reference root) whose behaviour it reproduces.
 *
 * Conventions
 *   - Complex samples are `yagi_cf32` = `num_complex::Complex<f32>` (#[repr(C)] {re, im}).
 *   - Type suffixes follow liquid-dsp / yagi's test names:
 *       rrrf = FirFilter<f32, f32>, crcf = FirFilter<Complex32, f32>,
 *       cccf = FirFilter<Complex32, Complex32>   (T = sample/output type, C = Coeff type).
 *   - Every function returns a yagi_status (0 = OK).  The variants mirror error::Error
 *     (src/error.rs:7-14); YAGI_ERR_DEVICE is added for HIP/RCCL failures.  Where the reference
 *     panics on a slice index (firdecim.rs:182, dotprod/mod.rs:104, fft/mod.rs:46) this library
 *     returns YAGI_ERR_CONFIG instead of aborting.  yagi_hip_last_error() gives the message
 *     (thread-local), like the String each Error variant carries.
 *   - Pointers are HOST pointers unless the function name ends in `_dev`, in which case data
 *     pointers are device (HBM) pointers and the call is asynchronous on the handle's stream.
 *   - NO ALIASING: the input and output ranges of a `_dev` block call must not overlap (no in-place
 *     execution): the kernels read tile halos and the carried window from the input while other
 *     workgroups already store the output.  Rust's borrows (`&[T]` in, `&mut [T]` out) make such a
 *     call unwritable in the reference; here it returns YAGI_ERR_CONFIG.
 *   - A handle owns its device taps, its device window (the filter state), a workspace and a
 *     stream reference.  One handle = one owner thread at a time (the reference's `&mut self`);
 *     distinct handles may be used concurrently.  Fft plans are read-only once created.
 *   - There is NO CPU fallback: every arithmetic result is produced by a HIP kernel; if no
 *     device is present the first call fails with YAGI_ERR_DEVICE.
 */
#ifndef YAGI_HIP_H
#define YAGI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } yagi_cf32;

typedef enum {
    YAGI_OK = 0,
    YAGI_ERR_INTERNAL = 1,        /* Error::Internal       error.rs:8  */
    YAGI_ERR_CONFIG = 2,          /* Error::Config         error.rs:9  */
    YAGI_ERR_VALUE = 3,           /* Error::Value          error.rs:10 */
    YAGI_ERR_RANGE = 4,           /* Error::Range          error.rs:11 */
    YAGI_ERR_MODE = 5,            /* Error::Mode           error.rs:12 */
    YAGI_ERR_NO_CONVERGENCE = 6,  /* Error::NoConvergence  error.rs:13 */
    YAGI_ERR_DEVICE = 7           /* HIP / RCCL failure (no reference counterpart) */
} yagi_status;

typedef void *yagi_stream_t;      /* a hipStream_t; NULL = the default stream */

/* fft::Direction (src/fft/mod.rs:13-17) */
#define YAGI_FFT_FORWARD 0
#define YAGI_FFT_BACKWARD 1

const char *yagi_hip_last_error(void);
const char *yagi_hip_version(void);

/* ---- device plumbing (no reference counterpart: the reference has no device boundary) ---- */
int yagi_hip_device_count(int *count);
int yagi_hip_set_device(int device);
int yagi_hip_malloc(void **dev_ptr, size_t bytes);
int yagi_hip_free(void *dev_ptr);
int yagi_hip_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int yagi_hip_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int yagi_hip_memset_dev(void *dst_dev, int value, size_t bytes);
int yagi_hip_device_synchronize(void);
int yagi_hip_stream_synchronize(yagi_stream_t stream);

/* Synthetic input: counter-based SplitMix64 + Box-Muller in the shape of
 * random::randnf / crandnf (src/random/normal.rs:9-44); sample i of a stream depends only on
 * (seed, first + i).  Complex samples have re, im ~ N(0, 1/2). */
int yagi_hip_gen_real_dev(uint64_t seed, uint64_t first, size_t n, float *x_dev, yagi_stream_t s);
int yagi_hip_gen_complex_dev(uint64_t seed, uint64_t first, size_t n, yagi_cf32 *x_dev, yagi_stream_t s);

/* ---- dotprod: trait DotProd (src/dotprod/mod.rs:13-73) -----------------------------------
 *   rrrf: [f32].[f32]            mod.rs:19-31      rccf: [f32].[Complex]      mod.rs:33-45
 *   crcf: [Complex].[f32]        mod.rs:47-59      cccf: [Complex].[Complex]  mod.rs:61-73
 * y = sum_i a[i]*b[i] over n = the common length.  Kernel: per-lane strided FMA, wave64
 * shuffle tree, LDS cross-wave, fixed-order two-pass combine (bitwise reproducible). */
int yagi_hip_dotprod_rrrf(const float *a, const float *b, size_t n, float *y);
int yagi_hip_dotprod_rccf(const float *a, const yagi_cf32 *b, size_t n, yagi_cf32 *y);
int yagi_hip_dotprod_crcf(const yagi_cf32 *a, const float *b, size_t n, yagi_cf32 *y);
int yagi_hip_dotprod_cccf(const yagi_cf32 *a, const yagi_cf32 *b, size_t n, yagi_cf32 *y);
int yagi_hip_dotprod_rrrf_dev(const float *a, const float *b, size_t n, float *y, yagi_stream_t s);
int yagi_hip_dotprod_rccf_dev(const float *a, const yagi_cf32 *b, size_t n, yagi_cf32 *y, yagi_stream_t s);
int yagi_hip_dotprod_crcf_dev(const yagi_cf32 *a, const float *b, size_t n, yagi_cf32 *y, yagi_stream_t s);
int yagi_hip_dotprod_cccf_dev(const yagi_cf32 *a, const yagi_cf32 *b, size_t n, yagi_cf32 *y, yagi_stream_t s);

/* ---- FIR objects --------------------------------------------------------------------------
 * YAGI_FIR_API(K, T, C) declares, for type combination K:
 *
 * firfilt  = FirFilter<T,C>             src/filter/fir/firfilt.rs
 *   create              new(h)                          :63-79   (h_len == 0 -> CONFIG)
 *   create_kaiser       new_kaiser(n, fc, as_, mu)      :93-97   (design: kaiser.rs:16-51)
 *   create_rect         new_rect(n)                     :149-155 (n in [1,1024])
 *   create_dc_blocker   new_dc_blocker(m, as_)          :166-170 (design/mod.rs:336-378 fir_design_notch at f0 = 0)
 *   create_notch        new_notch(m, as_, f0)           :183-186 (real taps: notch pair at +-f0; complex taps: the
 *                                                       DC blocker mixed to f0, firfilt.rs:25-44)
 *   clone               #[derive(Clone)]                :8       (state continues identically)
 *   set_coefficients    set_coefficients(h) (+reset)    :193-206
 *   reset               reset()                         :209-213
 *   push / write        push(x) / write(&[x])           :220-234
 *   execute             execute() -> T                  :241-246 (dotprod kernel on the window)
 *   execute_one         execute_one(x)                  :256-259
 *   execute_block       execute_block(x, y)             :267-278 (nx != ny -> CONFIG)
 *   set_scale/get_scale/get_length/get_coefficients      :285-314
 *   freqresponse        freqresponse(fc)                :325-328 (design/mod.rs:666-675; host arithmetic on the taps)
 *   groupdelay          groupdelay(fc)                  :339-342 (design/mod.rs:687-704)
 *   y[i] = scale * sum_{k<L} h[k] * x[i-k], history zero at start.
 *
 * firdecim = FirDecimationFilter<T,C>   src/filter/fir/firdecim.rs
 *   create              new(M, h, h_len)                :38-57   (h_len==0 or M==0 -> CONFIG)
 *   create_kaiser       new_kaiser(M, m, as_)           :70-87   (M<2, m==0, as_<0 -> CONFIG)
 *   execute             execute(&x[..M]) -> T           :179-191
 *   execute_block       execute_block(x, n, y)          :200-205 (x holds n*M samples)
 *   y[i] = scale * sum_k h[k] * x[i*M - k].
 *
 * firpfb   = FirPfbFilter<T,C>          src/filter/fir/firpfb.rs
 *   create              new(num_filters, h, h_len)      :34-65   (h_sub_len = h_len / num_filters)
 *   create_kaiser       new_kaiser(M, m, fc, as_)       :94-114
 *   create_default      default(M, m)                   :79-81
 *   push / write / execute(i) / execute_block(i, x, y)  :255-301 (i >= num_filters -> CONFIG)
 *   execute_all_dev     all num_filters branch outputs for every pushed sample (polyphase
 *                       interpolator form, firinterp.rs:224-231): y[n][i], n-major
 *   execute_select_dev  branch index per sample (the access pattern of resamp.rs:141-154)
 *   y_i[n] = scale * sum_k h[i + k*M] * x[n-k].
 *
 * `_dev` variants take device pointers and run asynchronously on the handle's stream.
 */
#define YAGI_FIR_API(K, T, C)                                                                       \
    typedef struct yagi_hip_firfilt_##K##_s *yagi_hip_firfilt_##K;                                  \
    int yagi_hip_firfilt_##K##_create(const C *h, size_t h_len, yagi_hip_firfilt_##K *q);           \
    int yagi_hip_firfilt_##K##_create_kaiser(size_t n, float fc, float as_, float mu,               \
                                             yagi_hip_firfilt_##K *q);                              \
    int yagi_hip_firfilt_##K##_create_rect(size_t n, yagi_hip_firfilt_##K *q);                      \
    int yagi_hip_firfilt_##K##_create_dc_blocker(size_t m, float as_, yagi_hip_firfilt_##K *q);     \
    int yagi_hip_firfilt_##K##_create_notch(size_t m, float as_, float f0, yagi_hip_firfilt_##K *q);\
    int yagi_hip_firfilt_##K##_destroy(yagi_hip_firfilt_##K q);                                     \
    int yagi_hip_firfilt_##K##_clone(yagi_hip_firfilt_##K q, yagi_hip_firfilt_##K *out);            \
    int yagi_hip_firfilt_##K##_set_stream(yagi_hip_firfilt_##K q, yagi_stream_t s);                 \
    int yagi_hip_firfilt_##K##_set_coefficients(yagi_hip_firfilt_##K q, const C *h, size_t h_len);  \
    int yagi_hip_firfilt_##K##_reset(yagi_hip_firfilt_##K q);                                       \
    int yagi_hip_firfilt_##K##_push(yagi_hip_firfilt_##K q, T x);                                   \
    int yagi_hip_firfilt_##K##_write(yagi_hip_firfilt_##K q, const T *x, size_t n);                 \
    int yagi_hip_firfilt_##K##_execute(yagi_hip_firfilt_##K q, T *y);                               \
    int yagi_hip_firfilt_##K##_execute_one(yagi_hip_firfilt_##K q, T x, T *y);                      \
    int yagi_hip_firfilt_##K##_execute_block(yagi_hip_firfilt_##K q, const T *x, size_t nx, T *y,   \
                                             size_t ny);                                            \
    int yagi_hip_firfilt_##K##_execute_block_dev(yagi_hip_firfilt_##K q, const T *x_dev, size_t n,  \
                                                 T *y_dev);                                         \
    int yagi_hip_firfilt_##K##_set_pipeline(yagi_hip_firfilt_##K q, int on);  /* as yagi_hip_firfft_crcf_set_pipeline: */ \
    int yagi_hip_firfilt_##K##_join(yagi_hip_firfilt_##K q);  /* consecutive execute_block_dev calls overlap; y_dev and x_dev's reuse are ordered after join */ \
    int yagi_hip_firfilt_##K##_set_scale(yagi_hip_firfilt_##K q, C scale);                          \
    int yagi_hip_firfilt_##K##_get_scale(yagi_hip_firfilt_##K q, C *scale);                         \
    int yagi_hip_firfilt_##K##_get_length(yagi_hip_firfilt_##K q, size_t *h_len);                   \
    int yagi_hip_firfilt_##K##_get_coefficients(yagi_hip_firfilt_##K q, C *h, size_t h_len);        \
    int yagi_hip_firfilt_##K##_freqresponse(yagi_hip_firfilt_##K q, float fc, yagi_cf32 *H);        \
    int yagi_hip_firfilt_##K##_groupdelay(yagi_hip_firfilt_##K q, float fc, float *delay);          \
                                                                                                    \
    typedef struct yagi_hip_firdecim_##K##_s *yagi_hip_firdecim_##K;                                \
    int yagi_hip_firdecim_##K##_create(size_t M, const C *h, size_t h_len,                          \
                                       yagi_hip_firdecim_##K *q);                                   \
    int yagi_hip_firdecim_##K##_create_kaiser(size_t M, size_t m, float as_,                        \
                                              yagi_hip_firdecim_##K *q);                            \
    int yagi_hip_firdecim_##K##_destroy(yagi_hip_firdecim_##K q);                                   \
    int yagi_hip_firdecim_##K##_clone(yagi_hip_firdecim_##K q, yagi_hip_firdecim_##K *out);         \
    int yagi_hip_firdecim_##K##_set_stream(yagi_hip_firdecim_##K q, yagi_stream_t s);               \
    int yagi_hip_firdecim_##K##_reset(yagi_hip_firdecim_##K q);                                     \
    int yagi_hip_firdecim_##K##_get_decim_rate(yagi_hip_firdecim_##K q, size_t *M);                 \
    int yagi_hip_firdecim_##K##_set_scale(yagi_hip_firdecim_##K q, C scale);                        \
    int yagi_hip_firdecim_##K##_get_scale(yagi_hip_firdecim_##K q, C *scale);                       \
    int yagi_hip_firdecim_##K##_freqresp(yagi_hip_firdecim_##K q, float fc, yagi_cf32 *H);          \
    int yagi_hip_firdecim_##K##_execute(yagi_hip_firdecim_##K q, const T *x, size_t nx, T *y);      \
    int yagi_hip_firdecim_##K##_execute_block(yagi_hip_firdecim_##K q, const T *x, size_t nx,       \
                                              size_t n, T *y);                                      \
    int yagi_hip_firdecim_##K##_execute_block_dev(yagi_hip_firdecim_##K q, const T *x_dev,          \
                                                  size_t n, T *y_dev);                              \
                                                                                                    \
    typedef struct yagi_hip_firpfb_##K##_s *yagi_hip_firpfb_##K;                                    \
    int yagi_hip_firpfb_##K##_create(size_t num_filters, const C *h, size_t h_len,                  \
                                     yagi_hip_firpfb_##K *q);                                       \
    int yagi_hip_firpfb_##K##_create_kaiser(size_t num_filters, size_t m, float fc, float as_,      \
                                            yagi_hip_firpfb_##K *q);                                \
    int yagi_hip_firpfb_##K##_create_default(size_t num_filters, size_t m,                          \
                                             yagi_hip_firpfb_##K *q);                               \
    int yagi_hip_firpfb_##K##_destroy(yagi_hip_firpfb_##K q);                                       \
    int yagi_hip_firpfb_##K##_clone(yagi_hip_firpfb_##K q, yagi_hip_firpfb_##K *out);               \
    int yagi_hip_firpfb_##K##_set_stream(yagi_hip_firpfb_##K q, yagi_stream_t s);                   \
    int yagi_hip_firpfb_##K##_reset(yagi_hip_firpfb_##K q);                                         \
    int yagi_hip_firpfb_##K##_set_scale(yagi_hip_firpfb_##K q, C scale);                            \
    int yagi_hip_firpfb_##K##_get_scale(yagi_hip_firpfb_##K q, C *scale);                           \
    int yagi_hip_firpfb_##K##_push(yagi_hip_firpfb_##K q, T x);                                     \
    int yagi_hip_firpfb_##K##_write(yagi_hip_firpfb_##K q, const T *x, size_t n);                   \
    int yagi_hip_firpfb_##K##_execute(yagi_hip_firpfb_##K q, size_t i, T *y);                       \
    int yagi_hip_firpfb_##K##_execute_block(yagi_hip_firpfb_##K q, size_t i, const T *x,            \
                                            size_t nx, T *y, size_t ny);                            \
    int yagi_hip_firpfb_##K##_execute_block_dev(yagi_hip_firpfb_##K q, size_t i, const T *x_dev,    \
                                                size_t n, T *y_dev);                                \
    int yagi_hip_firpfb_##K##_execute_all_dev(yagi_hip_firpfb_##K q, const T *x_dev, size_t n,      \
                                              T *y_dev);                                            \
    int yagi_hip_firpfb_##K##_execute_select_dev(yagi_hip_firpfb_##K q, const uint32_t *idx_dev,    \
                                                 const T *x_dev, size_t n, T *y_dev);

YAGI_FIR_API(rrrf, float, float)
YAGI_FIR_API(crcf, yagi_cf32, float)
YAGI_FIR_API(cccf, yagi_cf32, yagi_cf32)

/* ---- FftFilt<T,Coeff>: src/filter/fftfilt.rs (overlap-add fast convolution, block n, FFT 2n) ------
 *   create            create(h, n)                     :46-84   (h_len == 0 or n < h_len-1 -> CONFIG;
 *                                                               additionally 2n <= 8192 in this engine)
 *   reset             reset()                          :86-88
 *   set_scale/get_scale                                :95-101  (internally scale/(2n), like the reference)
 *   execute           execute(x, y)                    :103-138 (x.len() != n or y.len() != n -> CONFIG)
 *   execute_blocks    `nblocks` consecutive execute() calls as one batch (device-friendly form)
 *   get_length        get_length()                     :140-142
 * y = overlap-add of IFFT(FFT([x;0]) * FFT([h;0])) * scale; for rrrf the real part is returned
 * (FromComplex32 for f32, fftfilt.rs:16-20). */
#define YAGI_FFTFILT_API(K, T, C)                                                                   \
    typedef struct yagi_hip_fftfilt_##K##_s *yagi_hip_fftfilt_##K;                                  \
    int yagi_hip_fftfilt_##K##_create(const C *h, size_t h_len, size_t n, yagi_hip_fftfilt_##K *q); \
    int yagi_hip_fftfilt_##K##_destroy(yagi_hip_fftfilt_##K q);                                     \
    int yagi_hip_fftfilt_##K##_clone(yagi_hip_fftfilt_##K q, yagi_hip_fftfilt_##K *out);            \
    int yagi_hip_fftfilt_##K##_set_stream(yagi_hip_fftfilt_##K q, yagi_stream_t s);                 \
    int yagi_hip_fftfilt_##K##_reset(yagi_hip_fftfilt_##K q);                                       \
    int yagi_hip_fftfilt_##K##_set_scale(yagi_hip_fftfilt_##K q, C scale);                          \
    int yagi_hip_fftfilt_##K##_get_scale(yagi_hip_fftfilt_##K q, C *scale);                         \
    int yagi_hip_fftfilt_##K##_get_length(yagi_hip_fftfilt_##K q, size_t *h_len);                   \
    int yagi_hip_fftfilt_##K##_execute(yagi_hip_fftfilt_##K q, const T *x, size_t nx, T *y,         \
                                       size_t ny);                                                  \
    int yagi_hip_fftfilt_##K##_execute_blocks(yagi_hip_fftfilt_##K q, const T *x, size_t nblocks,   \
                                              T *y);                                                \
    int yagi_hip_fftfilt_##K##_execute_blocks_dev(yagi_hip_fftfilt_##K q, const T *x_dev,           \
                                                  size_t nblocks, T *y_dev);

YAGI_FFTFILT_API(rrrf, float, float)
YAGI_FFTFILT_API(crcf, yagi_cf32, float)
YAGI_FFTFILT_API(cccf, yagi_cf32, yagi_cf32)

/* ---- FirInterpolationFilter<T,Coeff>: src/filter/fir/firinterp.rs (polyphase interpolator) ---------
 *   create           new(interp, h, h_len)        :36-60   (interp < 2 or h_len < interp -> CONFIG; taps
 *                                                          zero-padded to a multiple of interp)
 *   create_kaiser    new_kaiser(interp, m, as_)   :73-90   (uses the first 2*interp*m taps)
 *   create_linear    new_linear(interp)           :135-147
 *   create_window    new_window(interp, m)        :159-174
 *   execute          execute(x, &mut y[..interp]) :224-231 (push x, then every branch in order)
 *   execute_block    execute_block(x, y)          :239-244 (y holds n*interp samples)
 *   flush            flush(y)                     :251-253 (execute with a zero input)
 *   reset / get_interp_rate / get_sub_len / set_scale / get_scale   :177-215
 * Device form: firpfb_all_kernel (all branches per pushed sample, output n-major). */
#define YAGI_FIRINTERP_API(K, T, C)                                                                 \
    typedef struct yagi_hip_firinterp_##K##_s *yagi_hip_firinterp_##K;                              \
    int yagi_hip_firinterp_##K##_create(size_t interp, const C *h, size_t h_len,                    \
                                        yagi_hip_firinterp_##K *q);                                 \
    int yagi_hip_firinterp_##K##_create_kaiser(size_t interp, size_t m, float as_,                  \
                                               yagi_hip_firinterp_##K *q);                          \
    int yagi_hip_firinterp_##K##_create_linear(size_t interp, yagi_hip_firinterp_##K *q);           \
    int yagi_hip_firinterp_##K##_create_window(size_t interp, size_t m, yagi_hip_firinterp_##K *q); \
    int yagi_hip_firinterp_##K##_destroy(yagi_hip_firinterp_##K q);                                 \
    int yagi_hip_firinterp_##K##_clone(yagi_hip_firinterp_##K q, yagi_hip_firinterp_##K *out);      \
    int yagi_hip_firinterp_##K##_set_stream(yagi_hip_firinterp_##K q, yagi_stream_t s);             \
    int yagi_hip_firinterp_##K##_reset(yagi_hip_firinterp_##K q);                                   \
    int yagi_hip_firinterp_##K##_get_interp_rate(yagi_hip_firinterp_##K q, size_t *interp);         \
    int yagi_hip_firinterp_##K##_get_sub_len(yagi_hip_firinterp_##K q, size_t *h_sub_len);          \
    int yagi_hip_firinterp_##K##_set_scale(yagi_hip_firinterp_##K q, C scale);                      \
    int yagi_hip_firinterp_##K##_get_scale(yagi_hip_firinterp_##K q, C *scale);                     \
    int yagi_hip_firinterp_##K##_execute(yagi_hip_firinterp_##K q, T x, T *y, size_t ny);           \
    int yagi_hip_firinterp_##K##_execute_block(yagi_hip_firinterp_##K q, const T *x, size_t nx,     \
                                               T *y, size_t ny);                                    \
    int yagi_hip_firinterp_##K##_execute_block_dev(yagi_hip_firinterp_##K q, const T *x_dev,        \
                                                   size_t n, T *y_dev);                             \
    int yagi_hip_firinterp_##K##_flush(yagi_hip_firinterp_##K q, T *y, size_t ny);

YAGI_FIRINTERP_API(rrrf, float, float)
YAGI_FIRINTERP_API(crcf, yagi_cf32, float)
YAGI_FIRINTERP_API(cccf, yagi_cf32, yagi_cf32)

/* ---- Rresamp<T,Coeff>: src/filter/resampler/rresamp.rs:8-183 (rational-rate resampler P/Q) -----
 *   create          new(interp, decim, m, h)            :28-57   bank = FirPfbFilter::new(interp, h, 2*interp*m)
 *   create_kaiser   new_kaiser(interp, decim, m, bw, as) :59-82   rates reduced by their gcd (block_len = gcd),
 *                                                                bw < 0 picks the default, scale 2 bw sqrt(Q/P)
 *   create_default  new_default(interp, decim)           :99-104  m 12, bw 0.5, 60 dB
 *   reset / set_scale / get_scale                        :106-116
 *   get_params      get_interp / get_decim / get_delay (= m) / get_block_len  :118-148
 *                   (get_rate = P/Q, get_p = P*block_len, get_q = Q*block_len follow from them)
 *   write           write(buf)                           :150-152 pushes samples without producing output
 *   execute         execute(x, y)                        :154-161 Q*block_len inputs -> P*block_len outputs
 *   execute_block   execute_block(x, n, y)               :163-170 n times execute.  (For block_len > 1 the
 *                   reference slices x and y by Q and P here and panics inside execute; this engine advances
 *                   by Q*block_len and P*block_len like liquid-dsp's rresamp_execute_block.)
 *   new_prototype (:84-97) needs fir_design_prototype (design code out of scope): create() with external taps.
 * Device form: rresamp_kernel -- the branch schedule is static (output n of a block = branch (nQ) mod P after
 * input floor(nQ/P)), so every output is an independent 2m-tap dot product. */
#define YAGI_RRESAMP_API(K, T, C)                                                                   \
    typedef struct yagi_hip_rresamp_##K##_s *yagi_hip_rresamp_##K;                                  \
    int yagi_hip_rresamp_##K##_create(size_t interp, size_t decim, size_t m, const C *h,            \
                                      size_t h_len, yagi_hip_rresamp_##K *q);                       \
    int yagi_hip_rresamp_##K##_create_kaiser(size_t interp, size_t decim, size_t m, float bw,       \
                                             float as_, yagi_hip_rresamp_##K *q);                   \
    int yagi_hip_rresamp_##K##_create_default(size_t interp, size_t decim, yagi_hip_rresamp_##K *q);\
    int yagi_hip_rresamp_##K##_destroy(yagi_hip_rresamp_##K q);                                     \
    int yagi_hip_rresamp_##K##_clone(yagi_hip_rresamp_##K q, yagi_hip_rresamp_##K *out); /* derive(Clone) :8 */ \
    int yagi_hip_rresamp_##K##_set_stream(yagi_hip_rresamp_##K q, yagi_stream_t s);                 \
    int yagi_hip_rresamp_##K##_reset(yagi_hip_rresamp_##K q);                                       \
    int yagi_hip_rresamp_##K##_set_scale(yagi_hip_rresamp_##K q, C scale);                          \
    int yagi_hip_rresamp_##K##_get_scale(yagi_hip_rresamp_##K q, C *scale);                         \
    int yagi_hip_rresamp_##K##_get_params(yagi_hip_rresamp_##K q, size_t *interp, size_t *decim,    \
                                          size_t *m, size_t *block_len);                            \
    int yagi_hip_rresamp_##K##_write(yagi_hip_rresamp_##K q, const T *x, size_t n);                 \
    int yagi_hip_rresamp_##K##_execute(yagi_hip_rresamp_##K q, const T *x, size_t nx, T *y,         \
                                       size_t ny);                                                  \
    int yagi_hip_rresamp_##K##_execute_block(yagi_hip_rresamp_##K q, const T *x, size_t nx,         \
                                             size_t n, T *y, size_t ny);                            \
    int yagi_hip_rresamp_##K##_execute_block_dev(yagi_hip_rresamp_##K q, const T *x_dev, size_t n,  \
                                                 T *y_dev);

YAGI_RRESAMP_API(rrrf, float, float)
YAGI_RRESAMP_API(crcf, yagi_cf32, float)
YAGI_RRESAMP_API(cccf, yagi_cf32, yagi_cf32)

/* ---- Resamp2<T,Coeff>: src/filter/resampler/resamp2.rs:26-180 (half-band filter / 2-channel bank / x2 resampler)
 *      MsResamp2<T,Coeff>: src/filter/resampler/msresamp2.rs:8-198 (2^S resampler = chain of half-band stages)
 *   create(hf, m, f0)        new(m, f0, as_) :44-88 FROM THE DESIGNED PROTOTYPE hf[4m+1]: the reference designs it with
 *                            fir_design_pm_halfband_stopband_attenuation (Parks-McClellan design code, outside the hot
 *                            path); everything after the design -- for_halfband modulation :9-23, the 2m branch taps
 *                            h1[i] = h[4m - 1 - 2i] :66-70, two 2m-sample windows, toggle -- is reproduced.
 *   create_kaiser(m, f0, as) the reference's signature with a Kaiser-windowed half-band prototype (kaiser(4m+1, 0.25, as))
 *   clone / reset / set_scale / get_scale / get_delay (= 2m - 1)                     :25,90-106
 *   execute_block[_dev](mode, x, nx, y, ny): nx input samples through one of the five forms, state carried across calls;
 *       ny = the length of y and must be the form's output count (YAGI_ERR_CONFIG otherwise: the reference's slices
 *       carry their lengths)
 *       mode 0 filter_execute       :108-130  nx samples  -> 2 nx outputs, (y0, y1) = (low, high) per sample
 *       mode 1 analyzer_execute     :132-143  nx/2 pairs  -> nx outputs,   (low, high) per pair
 *       mode 2 synthesizer_execute  :145-157  nx/2 pairs (low, high) -> nx outputs
 *       mode 3 decim_execute        :159-169  nx samples  -> nx/2 outputs
 *       mode 4 interp_execute       :171-180  nx samples  -> 2 nx outputs
 *     (modes 1-3 need an even nx.)  The per-call forms of the reference are these with nx = 1 or 2.
 *   msresamp2 create(interp, num_stages, fc, f0, as)  new() :38-93 with Kaiser half-band stages (stage plan :66-88:
 *                            estimate_req_filter_len, m = max(3, ceil((h_len - 1) / 4)), as + 5 dB)
 *   msresamp2 create_taps(interp, num_stages, m_stage, hf_all)  the same from externally designed stage prototypes
 *                            (hf_all = the stages' hf[4 m_s + 1] one after the other)
 *   msresamp2 execute_block[_dev](x, nx, y, ny)  n times execute() :137-152: interp nx = n -> ny = n 2^S, decim
 *                            nx = n 2^S -> ny = n (x 1/2^S); any other pair of lengths is YAGI_ERR_CONFIG (:181)
 *   msresamp2 get_params     get_type / get_num_stages / get_delay :95-135 (+ the stage semi-lengths) */
#define YAGI_RESAMP2_API(K, T, C)                                                                   \
    typedef struct yagi_hip_resamp2_##K##_s *yagi_hip_resamp2_##K;                                  \
    typedef struct yagi_hip_msresamp2_##K##_s *yagi_hip_msresamp2_##K;                              \
    int yagi_hip_resamp2_##K##_create(const float *hf, size_t m, float f0, yagi_hip_resamp2_##K *q);\
    int yagi_hip_resamp2_##K##_create_kaiser(size_t m, float f0, float as_, yagi_hip_resamp2_##K *q);\
    int yagi_hip_resamp2_##K##_destroy(yagi_hip_resamp2_##K q);                                     \
    int yagi_hip_resamp2_##K##_clone(yagi_hip_resamp2_##K q, yagi_hip_resamp2_##K *out);            \
    int yagi_hip_resamp2_##K##_reset(yagi_hip_resamp2_##K q);                                       \
    int yagi_hip_resamp2_##K##_set_stream(yagi_hip_resamp2_##K q, yagi_stream_t s);                 \
    int yagi_hip_resamp2_##K##_set_scale(yagi_hip_resamp2_##K q, C scale);                          \
    int yagi_hip_resamp2_##K##_get_scale(yagi_hip_resamp2_##K q, C *scale);                         \
    int yagi_hip_resamp2_##K##_get_delay(yagi_hip_resamp2_##K q, size_t *delay);                    \
    int yagi_hip_resamp2_##K##_execute_block(yagi_hip_resamp2_##K q, int mode, const T *x,          \
                                             size_t nx, T *y, size_t ny);                           \
    int yagi_hip_resamp2_##K##_execute_block_dev(yagi_hip_resamp2_##K q, int mode, const T *x_dev,  \
                                                 size_t nx, T *y_dev, size_t ny);                   \
    int yagi_hip_msresamp2_##K##_create(int interp, size_t num_stages, float fc, float f0,          \
                                        float as_, yagi_hip_msresamp2_##K *q);                      \
    int yagi_hip_msresamp2_##K##_create_taps(int interp, size_t num_stages, const size_t *m_stage,  \
                                             const float *hf_all, yagi_hip_msresamp2_##K *q);       \
    int yagi_hip_msresamp2_##K##_destroy(yagi_hip_msresamp2_##K q);                                 \
    int yagi_hip_msresamp2_##K##_clone(yagi_hip_msresamp2_##K q, yagi_hip_msresamp2_##K *out);      \
    int yagi_hip_msresamp2_##K##_reset(yagi_hip_msresamp2_##K q);                                   \
    int yagi_hip_msresamp2_##K##_set_stream(yagi_hip_msresamp2_##K q, yagi_stream_t s);             \
    int yagi_hip_msresamp2_##K##_get_params(yagi_hip_msresamp2_##K q, int *interp,                  \
                                            size_t *num_stages, float *delay, size_t *m_stage);     \
    int yagi_hip_msresamp2_##K##_execute_block(yagi_hip_msresamp2_##K q, const T *x, size_t nx,     \
                                               T *y, size_t ny);                                    \
    int yagi_hip_msresamp2_##K##_execute_block_dev(yagi_hip_msresamp2_##K q, const T *x_dev,        \
                                                   size_t nx, T *y_dev, size_t ny);

YAGI_RESAMP2_API(rrrf, float, float)
YAGI_RESAMP2_API(crcf, yagi_cf32, float)
YAGI_RESAMP2_API(cccf, yagi_cf32, yagi_cf32)

/* Which kernel execute_block uses.  0 = auto (always a direct form), 1 = general direct-form kernels
 * (fir_kernels.hip: register-window kernel for blocks >= 512 samples, interleaved-output kernel below; both add
 * the taps in the reference's order and never touch a tap past h_len, so a NaN poisons exactly h_len outputs),
 * 4 = overlap-save fast convolution (<= 2049 taps; stream_kernels.hip).
 * crcf also: 2 = hand-scheduled register-sliding direct form (<= 1024 taps; the crcf auto choice), 3 = MFMA
 * Toeplitz direct form (<= 256 taps); these two multiply zero-padded taps (a NaN's footprint is rounded up to
 * 32 taps).
 * The direct forms evaluate the reference's sums (exact on integer-valued data); the fast convolution agrees
 * with them to f32 rounding (rel. L2 <= 2e-6 against the f64 truth), like the reference's own FftFilt, and is
 * 2.5x (rrrf, crcf; 256 taps) to 5x (cccf, 256 taps) faster on long blocks, a tie for short filters (63 taps). */
int yagi_hip_firfilt_rrrf_set_kernel(yagi_hip_firfilt_rrrf q, int choice);
int yagi_hip_firfilt_crcf_set_kernel(yagi_hip_firfilt_crcf q, int choice);
int yagi_hip_firfilt_cccf_set_kernel(yagi_hip_firfilt_cccf q, int choice);

/* ---- Fft<f32>: src/fft/mod.rs:33-69 (arithmetic = rustfft 6.2 in the reference) ------------
 *   create       Fft::new(n, direction)        :39-43   any n >= 1 the engine supports
 *   run          run(input, output)            :45-48   out of place, unnormalised,
 *                                                       forward = e^{-j 2 pi n k / N}
 *   run_batch_dev  `batch` contiguous transforms, device pointers (config C3)
 *   shift        shift(input, n)               :50-57   swap halves (odd n: last stays)
 *   fft_run      fft_run(input, output, dir)   :66-69   plan + run in one call
 * Sizes: every n up to 2^23, powers of two up to 2^24.  Powers of two 256..8192: register kernels (one HBM
 * round trip); other n <= 8192: mixed-radix Stockham passes in LDS (register butterflies for 2/3/4/5/7/8/16,
 * direct sums for other primes up to 89); n > 8192 that splits as n1 n2 with both factors <= 8192 and 89-smooth
 * (every power of two, 10000, 48000, ...): four-step form (three transposes around the sub-transforms);
 * everything else -- a prime factor > 89 -- Bluestein's chirp-z form over a power of two.  Larger n returns CONFIG.  Plans with scratch (Bluestein,
 * four-step) are not safe for concurrent run() calls on one handle. */
typedef struct yagi_hip_fft_s *yagi_hip_fft;
int yagi_hip_fft_create(size_t n, int direction, yagi_hip_fft *plan);
int yagi_hip_fft_destroy(yagi_hip_fft plan);
int yagi_hip_fft_clone(yagi_hip_fft plan, yagi_hip_fft *out);
int yagi_hip_fft_len(yagi_hip_fft plan, size_t *n);
int yagi_hip_fft_run(yagi_hip_fft plan, const yagi_cf32 *input, size_t n_in, yagi_cf32 *output,
                     size_t n_out);
int yagi_hip_fft_run_batch_dev(yagi_hip_fft plan, const yagi_cf32 *in_dev, yagi_cf32 *out_dev,
                               size_t batch, yagi_stream_t s);
int yagi_hip_fft_shift(yagi_cf32 *buf, size_t n);
int yagi_hip_fft_shift_dev(yagi_cf32 *buf_dev, size_t n, size_t batch, yagi_stream_t s);
int yagi_hip_fft_run_oneshot(const yagi_cf32 *input, yagi_cf32 *output, size_t n, int direction);

/* ---- Spgram<T>: src/fft/spgram.rs (Welch spectral periodogram; the in-tree consumer of window -> FFT) ----
 *   spgramcf = Spgram<Complex32>, spgramf = Spgram<f32>.
 *   create          new(nfft, wtype, window_len, delay)   :49-125  (nfft < 2, window_len > nfft, window_len == 0,
 *                                                                   delay == 0, Kaiser with odd window_len -> CONFIG;
 *                                                                   additionally nfft <= 8192 in this engine)
 *   create_default  default(nfft) = new(nfft, Kaiser, nfft/2, nfft/4)   :128-131
 *   clear / reset                                          :135-155
 *   set_alpha / get_alpha   (-1 = accumulate, else [0,1])  :158-174,229
 *   set_freq / set_rate / get_nfft / get_window_len / get_delay / get_wtype / counters   :177-226
 *   push / write    one transform every `delay` samples    :237-259  (write = the batched device form)
 *   get_psd_mag / get_psd   fft-shifted linear / dB        :292-316  (like the reference, the linear scale
 *                                                                   factor is 0 unless alpha == -1)
 *   estimate_psd    one-shot                               :319-330
 * wtype uses the reference's WindowType discriminants (math/windows.rs:7-18). */
#define YAGI_WINDOW_HAMMING 1
#define YAGI_WINDOW_HANN 2
#define YAGI_WINDOW_BLACKMANHARRIS 3
#define YAGI_WINDOW_BLACKMANHARRIS7 4
#define YAGI_WINDOW_KAISER 5
#define YAGI_WINDOW_FLATTOP 6
#define YAGI_WINDOW_TRIANGULAR 7
#define YAGI_WINDOW_RCOSTAPER 8
#define YAGI_WINDOW_KBD 9
#define YAGI_SPGRAM_API(K, T)                                                                       \
    typedef struct yagi_hip_spgram##K##_s *yagi_hip_spgram##K;                                      \
    int yagi_hip_spgram##K##_create(size_t nfft, int wtype, size_t window_len, size_t delay,        \
                                    yagi_hip_spgram##K *q);                                         \
    int yagi_hip_spgram##K##_create_default(size_t nfft, yagi_hip_spgram##K *q);                    \
    int yagi_hip_spgram##K##_destroy(yagi_hip_spgram##K q);                                         \
    int yagi_hip_spgram##K##_set_stream(yagi_hip_spgram##K q, yagi_stream_t s);                     \
    int yagi_hip_spgram##K##_clear(yagi_hip_spgram##K q);                                           \
    int yagi_hip_spgram##K##_reset(yagi_hip_spgram##K q);                                           \
    int yagi_hip_spgram##K##_set_alpha(yagi_hip_spgram##K q, float alpha);                          \
    int yagi_hip_spgram##K##_get_alpha(yagi_hip_spgram##K q, float *alpha);                         \
    int yagi_hip_spgram##K##_set_freq(yagi_hip_spgram##K q, float freq);                            \
    int yagi_hip_spgram##K##_set_rate(yagi_hip_spgram##K q, float rate);                            \
    int yagi_hip_spgram##K##_get_params(yagi_hip_spgram##K q, size_t *nfft, size_t *window_len,     \
                                        size_t *delay, int *wtype);                                 \
    int yagi_hip_spgram##K##_get_counters(yagi_hip_spgram##K q, uint64_t *num_samples,              \
                                          uint64_t *num_samples_total, uint64_t *num_transforms,    \
                                          uint64_t *num_transforms_total);                          \
    int yagi_hip_spgram##K##_push(yagi_hip_spgram##K q, T x);                                       \
    int yagi_hip_spgram##K##_write(yagi_hip_spgram##K q, const T *x, size_t n);                     \
    int yagi_hip_spgram##K##_write_dev(yagi_hip_spgram##K q, const T *x_dev, size_t n);             \
    int yagi_hip_spgram##K##_get_psd_mag(yagi_hip_spgram##K q, float *psd, size_t n);               \
    int yagi_hip_spgram##K##_get_psd(yagi_hip_spgram##K q, float *psd, size_t n);                   \
    int yagi_hip_spgram##K##_estimate_psd(size_t nfft, const T *x, size_t n, float *psd);

YAGI_SPGRAM_API(cf, yagi_cf32)
YAGI_SPGRAM_API(f, float)

/* ---- the headline stream: FirFilter<Complex32,f32>::execute_block
 * (firfilt.rs:267-278) feeding consecutive nfft-sample frames to Fft::run forward
 * (fft/mod.rs:45-48).  1..2049 taps; nfft = 4096 is fused so the FIR output never touches HBM, any other nfft the
 * Fft object supports runs as overlap-save FIR + batched transform (two launches).  Filter state carries across
 * calls exactly like the FirFilter object's. */
typedef struct yagi_hip_firfft_crcf_s *yagi_hip_firfft_crcf;
int yagi_hip_firfft_crcf_create(const float *h, size_t h_len, size_t nfft, yagi_hip_firfft_crcf *q);
int yagi_hip_firfft_crcf_destroy(yagi_hip_firfft_crcf q);
int yagi_hip_firfft_crcf_set_stream(yagi_hip_firfft_crcf q, yagi_stream_t s);
int yagi_hip_firfft_crcf_set_scale(yagi_hip_firfft_crcf q, float scale);
int yagi_hip_firfft_crcf_reset(yagi_hip_firfft_crcf q);
/* variant: 0 = auto (4 up to 257 taps, 3 up to 2049 taps),
 *   1 = fused direct form, register-sliding VALU FIR (<= 1024 taps),
 *   2 = fused direct form, MFMA Toeplitz FIR (<= 256 taps),
 *   3 = fast convolution (overlap-save kernel, then batched FFT; <= 2049 taps),
 *   4 = frequency-domain filter, one launch: FFT{h}.FFT{frame} + FFT{frame-boundary correction} (<= 257 taps).
 * All variants carry the same filter state and agree to f32 rounding (rel. L2 <= 1e-5 per frame against the
 * f64 truth); only 1 and 2 evaluate the FIR sums themselves. */
int yagi_hip_firfft_crcf_set_variant(yagi_hip_firfft_crcf q, int variant);
int yagi_hip_firfft_crcf_execute(yagi_hip_firfft_crcf q, const yagi_cf32 *x, size_t nframes,
                                 yagi_cf32 *spectra);
int yagi_hip_firfft_crcf_execute_dev(yagi_hip_firfft_crcf q, const yagi_cf32 *x_dev,
                                     size_t nframes, yagi_cf32 *spectra_dev);
/* Pipelined block calls (frequency-domain form, nfft = 4096).  Consecutive blocks of the stream depend on each other
 * only through the L-sample filter window (firfilt.rs:13,220-223), and that window is the previous block's own last L
 * INPUT samples.  With the pipeline on, execute_dev runs consecutive blocks alternately on two streams owned by the
 * object, block b + 1 reading its window straight from the tail of block b's x_dev, so block b + 1 ramps up while
 * block b drains.  The contract changes in one point: x_dev must be complete on the object's stream when execute_dev
 * is called (as before), but spectra_dev -- and the right to overwrite or free x_dev -- is ordered on the object's
 * stream only after yagi_hip_firfft_crcf_join (stream waits + a 2 KiB window copy, no host synchronisation).  Every
 * other entry point of the object joins first; results are bit-identical to the unpipelined calls. */
int yagi_hip_firfft_crcf_set_pipeline(yagi_hip_firfft_crcf q, int on);
int yagi_hip_firfft_crcf_join(yagi_hip_firfft_crcf q);

/* ---- multichannel::firpfbch / firpfbch2 analyzers ------------------------------------------
 * ABSENT from the reference (src/multichannel/mod.rs is 0 lines; LIQUID_COMPAT.md:1765-1798).
 * Semantics are liquid-dsp's (the library yagi rewrites), composed from the reference's own
 * primitives: FirPfb-style branch split (firpfb.rs:45-52), Window (window.rs), dotprod, Fft.
 *   firpfbch  analyzer: M channels, p taps/branch, prototype h[0 .. p*M); one frame = M input
 *             samples -> M channel outputs: X[M-1-i] = branch_i . window_i ; y = DFT_M(X).
 *             create_kaiser(M, m, as_): h = kaiser(2*M*m+1, 0.5/M, as_), p = 2*m.
 *   firpfbch  synthesizer: one frame = M channel samples -> M output samples:
 *             v = IDFT_M(X) (unnormalised); branch i pushes v[i] and y[i] = branch_i . window_i, i.e.
 *             y[f M + i] = sum_n h[i + n M] v_{f-n}[i].  Own state, reset() clears both.
 *   firpfbch2 synthesizer: one step = M channel samples -> M/2 outputs; v_s = IDFT_M(X_s)/2, f = step parity, b = i + f M/2:
 *             y[s M/2 + i] = sum_n h[i + n M] v_{s-2n}[b] + sum_n h[i + M/2 + n M] v_{s-1-2n}[b].  With the analyzer's
 *             create_kaiser (cutoff 1/M) and create_kaiser_synthesizer (cutoff 0.5/M) prototypes the round trip
 *             reproduces the input delayed by 2 M m - M/2 + 1 samples (-60 dB at m = 3).
 *   firpfbch2 analyzer (2x oversampled): M even, branch length 2*m, h[0 .. 2*M*m); one step =
 *             M/2 inputs -> M outputs, alternating half rotation, y = IDFT_M(X)/M.
 *             create_kaiser(M, m, as_): h = kaiser(2*M*m+1, 1/M, as_) * M / sum(h).
 * Output layout [frame][channel].  The `_shard_dev` form computes only the sub-bands
 * k = rank + nranks*q (q < M/nranks) into yshard[step][q] so an 8-GPU node can all-gather them
 * (RCCL) -- see INTEGRATION.md; `assemble_dev` permutes the gathered [rank][step][q] slabs into
 * [step][channel]. */
typedef struct yagi_hip_firpfbch_crcf_s *yagi_hip_firpfbch_crcf;
int yagi_hip_firpfbch_crcf_create(size_t M, size_t p, const float *h, yagi_hip_firpfbch_crcf *q);
int yagi_hip_firpfbch_crcf_create_kaiser(size_t M, size_t m, float as_, yagi_hip_firpfbch_crcf *q);
int yagi_hip_firpfbch_crcf_destroy(yagi_hip_firpfbch_crcf q);
int yagi_hip_firpfbch_crcf_set_stream(yagi_hip_firpfbch_crcf q, yagi_stream_t s);
int yagi_hip_firpfbch_crcf_reset(yagi_hip_firpfbch_crcf q);
int yagi_hip_firpfbch_crcf_analyzer_execute(yagi_hip_firpfbch_crcf q, const yagi_cf32 *x,
                                            size_t nframes, yagi_cf32 *y);
int yagi_hip_firpfbch_crcf_analyzer_execute_dev(yagi_hip_firpfbch_crcf q, const yagi_cf32 *x_dev,
                                                size_t nframes, yagi_cf32 *y_dev);
int yagi_hip_firpfbch_crcf_synthesizer_execute(yagi_hip_firpfbch_crcf q, const yagi_cf32 *x,
                                               size_t nframes, yagi_cf32 *y);
int yagi_hip_firpfbch_crcf_synthesizer_execute_dev(yagi_hip_firpfbch_crcf q, const yagi_cf32 *x_dev,
                                                   size_t nframes, yagi_cf32 *y_dev);

typedef struct yagi_hip_firpfbch2_crcf_s *yagi_hip_firpfbch2_crcf;
int yagi_hip_firpfbch2_crcf_create(size_t M, size_t m, const float *h, yagi_hip_firpfbch2_crcf *q);
int yagi_hip_firpfbch2_crcf_create_kaiser(size_t M, size_t m, float as_, yagi_hip_firpfbch2_crcf *q);
/* prototype of the matching synthesizer: kaiser(2*M*m+1, 0.5/M, as_) * M / sum(h) */
int yagi_hip_firpfbch2_crcf_create_kaiser_synthesizer(size_t M, size_t m, float as_,
                                                      yagi_hip_firpfbch2_crcf *q);
int yagi_hip_firpfbch2_crcf_destroy(yagi_hip_firpfbch2_crcf q);
int yagi_hip_firpfbch2_crcf_set_stream(yagi_hip_firpfbch2_crcf q, yagi_stream_t s);
int yagi_hip_firpfbch2_crcf_reset(yagi_hip_firpfbch2_crcf q);
int yagi_hip_firpfbch2_crcf_analyzer_execute(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x,
                                             size_t nsteps, yagi_cf32 *y);
int yagi_hip_firpfbch2_crcf_analyzer_execute_dev(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x_dev,
                                                 size_t nsteps, yagi_cf32 *y_dev);
/* synthesizer: one step = M channel samples -> M/2 output samples (include note above); own state and step parity */
int yagi_hip_firpfbch2_crcf_synthesizer_execute(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x,
                                                size_t nsteps, yagi_cf32 *y);
int yagi_hip_firpfbch2_crcf_synthesizer_execute_dev(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x_dev,
                                                    size_t nsteps, yagi_cf32 *y_dev);
int yagi_hip_firpfbch2_crcf_analyzer_execute_shard_dev(yagi_hip_firpfbch2_crcf q,
                                                       const yagi_cf32 *x_dev, size_t nsteps,
                                                       int rank, int nranks, yagi_cf32 *yshard_dev);
int yagi_hip_firpfbch2_crcf_assemble_dev(const yagi_cf32 *gathered_dev, size_t nsteps, size_t M,
                                         int nranks, yagi_cf32 *y_dev, yagi_stream_t s);

/* ---- multi-GPU: RCCL over xGMI (one process per GPU) ------------------------------------------
 * The reference has no distributed layer; the one exchange step of the hot path is the
 * firpfbch2 analyzer with its output sub-bands sharded over the GPUs of a node.
 * A communicator is created the RCCL way: ONE rank calls comm_unique_id and hands the 128 bytes to every rank by
 * whatever the host has (MPI, a socket, torch.distributed); every rank then calls comm_create (collective) with
 * its GPU current (yagi_hip_set_device).  RCCL is bound at first use (dlopen of librccl.so.1); without it these
 * calls return YAGI_ERR_DEVICE and every other entry point of the library works unchanged.
 *
 * analyzer_execute_sharded_dev: rank r of R computes the sub-bands k = r + R q of `nsteps` steps (chunk by chunk),
 * the chunks are all-gathered on the communicator's own stream while the object's stream runs the next chunk's
 * kernel, and a permutation kernel assembles y[step][channel] -- identical on every rank, and identical to
 * analyzer_execute_dev on one GPU.  nchunks = 0 picks a default (8, fewer for short blocks); nchunks < 0 forces the
 * sharded pipeline (|nchunks| chunks) even for a one-rank communicator (tests).  x, y are device pointers; the
 * call is asynchronous: on return the object's stream waits for the last chunk's assembly. */
#define YAGI_HIP_COMM_ID_BYTES 128
typedef struct yagi_hip_comm_s *yagi_hip_comm;
int yagi_hip_comm_unique_id(unsigned char *id /* [YAGI_HIP_COMM_ID_BYTES] */);
int yagi_hip_comm_create(const unsigned char *id, int rank, int nranks, yagi_hip_comm *comm);
int yagi_hip_comm_destroy(yagi_hip_comm comm);
int yagi_hip_comm_rank(yagi_hip_comm comm, int *rank, int *nranks);
int yagi_hip_comm_all_gather_dev(yagi_hip_comm comm, const void *send_dev, void *recv_dev,
                                 size_t bytes_per_rank, yagi_stream_t s);
int yagi_hip_firpfbch2_crcf_analyzer_execute_sharded_dev(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x_dev,
                                                         size_t nsteps, yagi_hip_comm comm, int nchunks,
                                                         yagi_cf32 *y_dev);

/* ---- design helper exposed for hosts that want the taps (kaiser.rs:16-51) ---------------- */
int yagi_hip_fir_design_kaiser(size_t n, float fc, float as_, float mu, float *h);

#ifdef __cplusplus
}
#endif
#endif /* YAGI_HIP_H */
