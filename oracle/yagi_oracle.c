/*
 * yagi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C CPU restatement of the yagi (EEGKit/yagi, Rust) algorithms on the FIR/FFT hot
 * path.  It exists to CHECK the HIP kernels (tests/, __graft_entry__.smoke()) and to be
 * timed as the "cpu_baseline" leg of bench.py.  Nothing under yagi_amd/ may import, link
 * or call it; the product path fails loudly when the HIP library is missing.
 *
 * Parity status:
 *   - dotprod / firfilt / firdecim / firpfb / Window: restated line by line from the
 *     reference and PINNED by the reference's own golden vectors (the .npz fixtures under tests/golden,
 *     tests/test_oracle_golden.py).
 *   - fft: the reference delegates the arithmetic to the third-party crate
 *     rustfft = "6.2.0" (Cargo.toml:21; Cargo.lock is git-ignored), whose source is not
 *     under /root/reference.  The oracle is therefore the *definition* the reference's
 *     tests pin (unnormalised DFT, forward = e^{-j2pi nk/N}): an f64 O(N^2) DFT, checked
 *     against all 33 FFT_TEST_{X,Y}N golden pairs (N <= 509).  At N = 4096 the reference
 *     holds no vector: PARITY UNPINNED by the reference there, pinned by definition only.
 *   - firpfbch / firpfbch2: the reference module is empty (src/multichannel/mod.rs, 0
 *     lines).  PARITY UNPINNED; the restatement below composes the reference's own
 *     FirPfb-style branch split + Window + dotprod + Fft following liquid-dsp's published
 *     firpfbch.c / firpfbch2.c semantics and is pinned by property tests only.
 *   - The reference is Rust and no Rust toolchain exists here: there is no oracle/_ref.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (Rust never contracts a*b+c into an FMA
 * and never reassociates), see oracle/Makefile.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } cf32;   /* num_complex::Complex<f32>, #[repr(C)] */
typedef struct { double re, im; } cf64;

#define YO_OK 0
#define YO_ECONFIG 2
#define YO_ERANGE 4

/* ------------------------------------------------------------------------------------
 * scalar arithmetic exactly as num-complex spells it (no FMA, no reassociation)
 * ---------------------------------------------------------------------------------- */
static inline cf32 c_add(cf32 a, cf32 b) { cf32 r = { a.re + b.re, a.im + b.im }; return r; }
/* Complex<f32> * f32  and  f32 * Complex<f32>  (component-wise, commutative bit for bit) */
static inline cf32 c_mulr(cf32 a, float b) { cf32 r = { a.re * b, a.im * b }; return r; }
/* Complex * Complex: (ar*br - ai*bi, ar*bi + ai*br) */
static inline cf32 c_mul(cf32 a, cf32 b) {
    cf32 r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}

/* ------------------------------------------------------------------------------------
 * dotprod  --  src/dotprod/mod.rs:19-73 (default build: iter().zip().map(a*b).sum(),
 * i.e. a strictly sequential left-to-right f32 accumulation starting from zero)
 * ---------------------------------------------------------------------------------- */
float yo_dotprod_rrrf(const float *a, const float *b, size_t n) {          /* :19-31 */
    float s = 0.0f;
    for (size_t i = 0; i < n; i++) s = s + a[i] * b[i];
    return s;
}
void yo_dotprod_rcc(const float *a, const cf32 *b, size_t n, cf32 *y) {    /* :33-45 */
    cf32 s = { 0.0f, 0.0f };
    for (size_t i = 0; i < n; i++) s = c_add(s, c_mulr(b[i], a[i]));
    *y = s;
}
void yo_dotprod_crc(const cf32 *a, const float *b, size_t n, cf32 *y) {    /* :47-59 */
    cf32 s = { 0.0f, 0.0f };
    for (size_t i = 0; i < n; i++) s = c_add(s, c_mulr(a[i], b[i]));
    *y = s;
}
void yo_dotprod_ccc(const cf32 *a, const cf32 *b, size_t n, cf32 *y) {     /* :61-73 */
    cf32 s = { 0.0f, 0.0f };
    for (size_t i = 0; i < n; i++) s = c_add(s, c_mul(a[i], b[i]));
    *y = s;
}
/* f64 truth for the same inner products (products and sums in double) */
double yo_dotprod_rrrf_f64(const float *a, const float *b, size_t n) {
    double s = 0.0;
    for (size_t i = 0; i < n; i++) s += (double)a[i] * (double)b[i];
    return s;
}
void yo_dotprod_crc_f64(const cf32 *a, const float *b, size_t n, cf64 *y) {
    double sr = 0, si = 0;
    for (size_t i = 0; i < n; i++) { sr += (double)a[i].re * b[i]; si += (double)a[i].im * b[i]; }
    y->re = sr; y->im = si;
}
void yo_dotprod_ccc_f64(const cf32 *a, const cf32 *b, size_t n, cf64 *y) {
    double sr = 0, si = 0;
    for (size_t i = 0; i < n; i++) {
        sr += (double)a[i].re * b[i].re - (double)a[i].im * b[i].im;
        si += (double)a[i].re * b[i].im + (double)a[i].im * b[i].re;
    }
    y->re = sr; y->im = si;
}

/* ------------------------------------------------------------------------------------
 * Window<T>  --  src/buffer/window.rs:4-92 ; msb_index src/utility/bits.rs:110-112
 * (element size generic: elements are `esz` bytes, copied with memcpy)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    unsigned char *v;
    size_t esz, len, n, mask, read_index, num_allocated;
} yo_window;

static unsigned msb_index(uint32_t x) {                /* bits.rs:110-112 */
    unsigned lz = 0;
    if (x == 0) return 0;
    while (!(x & 0x80000000u)) { x <<= 1; lz++; }
    return 32 - lz;
}

yo_window *yo_window_create(size_t n, size_t esz) {     /* window.rs:13-34 */
    if (n == 0) return NULL;                            /* Error::Config */
    yo_window *w = (yo_window *)calloc(1, sizeof(*w));
    unsigned m = msb_index((uint32_t)n);
    w->esz = esz; w->len = n; w->n = (size_t)1 << m; w->mask = w->n - 1;
    w->num_allocated = w->n + n - 1;
    w->v = (unsigned char *)calloc(w->num_allocated, esz);
    w->read_index = 0;
    return w;
}
void yo_window_destroy(yo_window *w) { if (w) { free(w->v); free(w); } }
void yo_window_reset(yo_window *w) {                    /* :61-64 */
    w->read_index = 0;
    memset(w->v, 0, w->num_allocated * w->esz);
}
const void *yo_window_read(const yo_window *w) {        /* :66-68 contiguous, OLDEST first */
    return w->v + w->read_index * w->esz;
}
int yo_window_index(const yo_window *w, size_t i, void *out) {   /* :70-75 */
    if (i >= w->len) return YO_ERANGE;
    memcpy(out, w->v + (w->read_index + i) * w->esz, w->esz);
    return YO_OK;
}
void yo_window_push(yo_window *w, const void *value) {  /* :77-85 */
    w->read_index = (w->read_index + 1) & w->mask;
    if (w->read_index == 0)
        memmove(w->v, w->v + w->n * w->esz, (w->len - 1) * w->esz);
    memcpy(w->v + (w->read_index + w->len - 1) * w->esz, value, w->esz);
}
void yo_window_write(yo_window *w, const void *values, size_t count) {   /* :87-91 */
    const unsigned char *p = (const unsigned char *)values;
    for (size_t i = 0; i < count; i++) yo_window_push(w, p + i * w->esz);
}
yo_window *yo_window_clone(const yo_window *s) {        /* #[derive(Clone)] :3 */
    yo_window *w = (yo_window *)malloc(sizeof(*w));
    *w = *s;
    w->v = (unsigned char *)malloc(s->num_allocated * s->esz);
    memcpy(w->v, s->v, s->num_allocated * s->esz);
    return w;
}
int yo_window_resize(yo_window *w, size_t n) {          /* :36-59 */
    if (n == w->len) return YO_OK;
    yo_window *nw = yo_window_create(n, w->esz);
    if (!nw) return YO_ECONFIG;
    unsigned char tmp[16] = { 0 }, val[16];
    if (n > w->len) {
        for (size_t i = 0; i < n - w->len; i++) yo_window_push(nw, tmp);
        for (size_t i = 0; i < w->len; i++) { yo_window_index(w, i, val); yo_window_push(nw, val); }
    } else {
        for (size_t i = w->len - n; i < w->len; i++) { yo_window_index(w, i, val); yo_window_push(nw, val); }
    }
    free(w->v);
    *w = *nw;
    free(nw);
    return YO_OK;
}

/* ------------------------------------------------------------------------------------
 * FirFilter / FirDecimationFilter / FirPfbFilter for the three type combinations.
 * One macro instantiates the restatement for (T, Coeff) = (f32,f32) "rrrf",
 * (Complex32,f32) "crcf", (Complex32,Complex32) "cccf".
 *
 * FirFilter state is a VecDeque<T> created full (firfilt.rs:72), so push() =
 * rotate_right(1) + w[0]=x (firfilt.rs:220-223) just moves `head` back by one and
 * overwrites the oldest slot; as_slices() (dotprod/mod.rs:99-108) splits the logical
 * sequence at cap-head.  The ring below reproduces that physical layout so the two
 * partial sums associate exactly as the reference's do.
 * ---------------------------------------------------------------------------------- */
#define DEFINE_FIR(SUF, T, C, MUL_TC, ADD_T, ZERO_T, ONE_C)                                   \
typedef struct { C *h; size_t h_len; T *w; size_t head; C scale; } yo_firfilt_##SUF;          \
                                                                                              \
yo_firfilt_##SUF *yo_firfilt_##SUF##_create(const C *h, size_t h_len) { /* firfilt.rs:63-79 */\
    if (h_len == 0) return NULL;                                                              \
    yo_firfilt_##SUF *q = (yo_firfilt_##SUF *)calloc(1, sizeof(*q));                          \
    q->h = (C *)malloc(h_len * sizeof(C)); memcpy(q->h, h, h_len * sizeof(C));                \
    q->h_len = h_len; q->w = (T *)calloc(h_len, sizeof(T)); q->head = 0; q->scale = ONE_C;    \
    return q;                                                                                 \
}                                                                                             \
void yo_firfilt_##SUF##_destroy(yo_firfilt_##SUF *q) { if (q) { free(q->h); free(q->w); free(q); } } \
yo_firfilt_##SUF *yo_firfilt_##SUF##_clone(const yo_firfilt_##SUF *s) { /* derive(Clone) :8 */\
    yo_firfilt_##SUF *q = yo_firfilt_##SUF##_create(s->h, s->h_len);                          \
    memcpy(q->w, s->w, s->h_len * sizeof(T)); q->head = s->head; q->scale = s->scale;         \
    return q;                                                                                 \
}                                                                                             \
void yo_firfilt_##SUF##_reset(yo_firfilt_##SUF *q) {                  /* :209-213 */          \
    /* reset() zeroes elements in place; the ring position is kept */                         \
    for (size_t i = 0; i < q->h_len; i++) q->w[i] = ZERO_T;                                   \
}                                                                                             \
void yo_firfilt_##SUF##_set_scale(yo_firfilt_##SUF *q, C s) { q->scale = s; } /* :285 */      \
void yo_firfilt_##SUF##_push(yo_firfilt_##SUF *q, T x) {              /* :220-223 */          \
    q->head = (q->head == 0) ? q->h_len - 1 : q->head - 1;                                    \
    q->w[q->head] = x;                                                                        \
}                                                                                             \
T yo_firfilt_##SUF##_execute(const yo_firfilt_##SUF *q) {             /* :241-246 */          \
    size_t split = q->h_len - q->head;      /* l = buf[head..cap], r = buf[0..head] */        \
    T l = ZERO_T, r = ZERO_T;                                                                 \
    for (size_t k = 0; k < split; k++) l = ADD_T(l, MUL_TC(q->w[q->head + k], q->h[k]));      \
    for (size_t k = split; k < q->h_len; k++) r = ADD_T(r, MUL_TC(q->w[k - split], q->h[k])); \
    return MUL_TC(ADD_T(l, r), q->scale);                                                     \
}                                                                                             \
int yo_firfilt_##SUF##_execute_block(yo_firfilt_##SUF *q, const T *x, size_t nx,              \
                                     T *y, size_t ny) {               /* :267-278 */          \
    if (nx != ny) return YO_ECONFIG;                                                          \
    for (size_t i = 0; i < nx; i++) { yo_firfilt_##SUF##_push(q, x[i]); y[i] = yo_firfilt_##SUF##_execute(q); } \
    return YO_OK;                                                                             \
}                                                                                             \
                                                                                              \
/* ---- FirDecimationFilter: firdecim.rs:12-17,38-57,179-205 ---- */                          \
typedef struct { C *h; size_t h_len, M; yo_window *w; C scale; } yo_firdecim_##SUF;           \
yo_firdecim_##SUF *yo_firdecim_##SUF##_create(size_t M, const C *h, size_t h_len) {           \
    if (h_len == 0 || M == 0) return NULL;                             /* :39-44 */           \
    yo_firdecim_##SUF *q = (yo_firdecim_##SUF *)calloc(1, sizeof(*q));                        \
    q->h = (C *)malloc(h_len * sizeof(C));                                                    \
    for (size_t i = 0; i < h_len; i++) q->h[i] = h[h_len - 1 - i];     /* :47 reversed */     \
    q->h_len = h_len; q->M = M; q->w = yo_window_create(h_len, sizeof(T)); q->scale = ONE_C;  \
    return q;                                                                                 \
}                                                                                             \
void yo_firdecim_##SUF##_destroy(yo_firdecim_##SUF *q) { if (q) { free(q->h); yo_window_destroy(q->w); free(q); } } \
void yo_firdecim_##SUF##_reset(yo_firdecim_##SUF *q) { yo_window_reset(q->w); }               \
void yo_firdecim_##SUF##_set_scale(yo_firdecim_##SUF *q, C s) { q->scale = s; }               \
T yo_firdecim_##SUF##_execute(yo_firdecim_##SUF *q, const T *x) {      /* :179-191 */         \
    T y = ZERO_T;                                                                             \
    for (size_t i = 0; i < q->M; i++) {                                                       \
        yo_window_push(q->w, &x[i]);                                                          \
        if (i == 0) {                                                                         \
            const T *r = (const T *)yo_window_read(q->w);                                     \
            T s = ZERO_T;                          /* [Coeff]·[T]: sum of h[k]*r[k] */         \
            for (size_t k = 0; k < q->h_len; k++) s = ADD_T(s, MUL_TC(r[k], q->h[k]));        \
            y = MUL_TC(s, q->scale);                                                          \
        }                                                                                     \
    }                                                                                         \
    return y;                                                                                 \
}                                                                                             \
void yo_firdecim_##SUF##_execute_block(yo_firdecim_##SUF *q, const T *x, size_t n, T *y) {    \
    for (size_t i = 0; i < n; i++) y[i] = yo_firdecim_##SUF##_execute(q, x + i * q->M); /* :200-205 */ \
}                                                                                             \
                                                                                              \
/* ---- FirPfbFilter: firpfb.rs:10-15,34-65,255-301 ---- */                                   \
typedef struct { size_t num_filters, h_sub_len; yo_window *w; C *filters; C scale; } yo_firpfb_##SUF; \
yo_firpfb_##SUF *yo_firpfb_##SUF##_create(size_t num_filters, const C *h, size_t h_len) {     \
    if (num_filters == 0 || h_len == 0) return NULL;                   /* :35-40 */           \
    size_t hs = h_len / num_filters;                                   /* :42 floor */        \
    if (hs == 0) return NULL;                       /* Window::new(0) -> Err (window.rs:14) */\
    yo_firpfb_##SUF *q = (yo_firpfb_##SUF *)calloc(1, sizeof(*q));                            \
    q->num_filters = num_filters; q->h_sub_len = hs;                                          \
    q->filters = (C *)malloc(num_filters * hs * sizeof(C));                                   \
    for (size_t i = 0; i < num_filters; i++)                                                  \
        for (size_t n = 0; n < hs; n++)                                                       \
            q->filters[i * hs + (hs - n - 1)] = h[i + n * num_filters];  /* :45-52 */         \
    q->w = yo_window_create(hs, sizeof(T)); q->scale = ONE_C;                                 \
    return q;                                                                                 \
}                                                                                             \
void yo_firpfb_##SUF##_destroy(yo_firpfb_##SUF *q) { if (q) { free(q->filters); yo_window_destroy(q->w); free(q); } } \
void yo_firpfb_##SUF##_reset(yo_firpfb_##SUF *q) { yo_window_reset(q->w); }                   \
void yo_firpfb_##SUF##_set_scale(yo_firpfb_##SUF *q, C s) { q->scale = s; }                   \
void yo_firpfb_##SUF##_push(yo_firpfb_##SUF *q, T x) { yo_window_push(q->w, &x); } /* :255 */ \
int yo_firpfb_##SUF##_execute(yo_firpfb_##SUF *q, size_t i, T *y) {    /* :277-286 */         \
    if (i >= q->num_filters) return YO_ECONFIG;                                               \
    const T *r = (const T *)yo_window_read(q->w);                                             \
    const C *f = q->filters + i * q->h_sub_len;                                               \
    T s = ZERO_T;                                                                             \
    for (size_t k = 0; k < q->h_sub_len; k++) s = ADD_T(s, MUL_TC(r[k], f[k]));               \
    *y = MUL_TC(s, q->scale);                                                                 \
    return YO_OK;                                                                             \
}                                                                                             \
int yo_firpfb_##SUF##_execute_block(yo_firpfb_##SUF *q, size_t i, const T *x, size_t n, T *y) { /* :295-301 */ \
    for (size_t k = 0; k < n; k++) {                                                          \
        yo_firpfb_##SUF##_push(q, x[k]);                                                      \
        int rc = yo_firpfb_##SUF##_execute(q, i, &y[k]);                                      \
        if (rc) return rc;                                                                    \
    }                                                                                         \
    return YO_OK;                                                                             \
}

static inline float r_add(float a, float b) { return a + b; }
static inline float r_mul(float a, float b) { return a * b; }
static const cf32 CZERO = { 0.0f, 0.0f };
static const cf32 CONE = { 1.0f, 0.0f };

DEFINE_FIR(rrrf, float, float, r_mul, r_add, 0.0f, 1.0f)
DEFINE_FIR(crcf, cf32, float, c_mulr, c_add, CZERO, 1.0f)
DEFINE_FIR(cccf, cf32, cf32, c_mul, c_add, CZERO, CONE)

/* ------------------------------------------------------------------------------------
 * f64 truth for the whole FIR family: y[i] = scale * sum_k h[k] * x[i*M + phase - k],
 * zero history.  kind: 0 rrrf, 1 crcf, 2 cccf.
 * x has nx samples; y gets n outputs; decimation M >= 1; out is interleaved double
 * (1 double per output for rrrf, 2 otherwise).
 * ---------------------------------------------------------------------------------- */
void yo_fir_block_f64(int kind, const float *h, size_t L, const float *scale,
                      const float *x, size_t nx, size_t M, size_t n, double *y) {
    for (size_t i = 0; i < n; i++) {
        size_t c = i * M;                     /* newest input index feeding this output */
        double sr = 0.0, si = 0.0;
        for (size_t k = 0; k < L && k <= c; k++) {
            size_t j = c - k;
            if (j >= nx) continue;
            if (kind == 0) sr += (double)h[k] * x[j];
            else if (kind == 1) { sr += (double)h[k] * x[2 * j]; si += (double)h[k] * x[2 * j + 1]; }
            else {
                double hr = h[2 * k], hi = h[2 * k + 1], xr = x[2 * j], xi = x[2 * j + 1];
                sr += hr * xr - hi * xi; si += hr * xi + hi * xr;
            }
        }
        if (kind == 0) y[i] = sr * scale[0];
        else if (kind == 1) { y[2 * i] = sr * scale[0]; y[2 * i + 1] = si * scale[0]; }
        else {
            double cr = scale[0], ci = scale[1];
            y[2 * i] = sr * cr - si * ci; y[2 * i + 1] = sr * ci + si * cr;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * FFT.  src/fft/mod.rs:13-26,39-57: unnormalised, Forward = e^{-j 2 pi n k / N},
 * Backward = e^{+...}; run() is out of place; shift() swaps halves (odd n: the last
 * element stays).  dir: 0 forward, 1 backward.
 * ---------------------------------------------------------------------------------- */
void yo_dft_f64(const cf32 *x, size_t n, int dir, cf64 *y) {   /* the definition; O(n^2) */
    cf64 *tw = (cf64 *)malloc(n * sizeof(cf64));
    const double s = dir ? 1.0 : -1.0;
    for (size_t m = 0; m < n; m++) {
        double a = s * 2.0 * M_PI * (double)m / (double)n;
        tw[m].re = cos(a); tw[m].im = sin(a);
    }
    for (size_t k = 0; k < n; k++) {
        double sr = 0, si = 0;
        size_t m = 0;
        for (size_t j = 0; j < n; j++) {
            sr += x[j].re * tw[m].re - x[j].im * tw[m].im;
            si += x[j].re * tw[m].im + x[j].im * tw[m].re;
            m += k; if (m >= n) m -= n;
        }
        y[k].re = sr; y[k].im = si;
    }
    free(tw);
}

void yo_fft_shift(cf32 *v, size_t n) {                         /* fft/mod.rs:50-57 */
    size_t n2 = (n % 2 == 0) ? n / 2 : (n - 1) / 2;
    for (size_t i = 0; i < n2; i++) { cf32 t = v[i]; v[i] = v[i + n2]; v[i + n2] = t; }
}

/* f32 power-of-two FFT used as the timed CPU baseline for 4096-point transforms: an
 * iterative decimation-in-time radix-4 (radix-2 first pass when log2 n is odd), the
 * published structure of rustfft's scalar Radix4 plan (what FftPlanner picks for 2^k
 * without SIMD).  Plan = digit-reversal table + twiddles computed in f64. */
typedef struct { size_t n; int dir; uint32_t *rev; cf32 *tw; } yo_fft_plan;

yo_fft_plan *yo_fft_plan_create(size_t n, int dir) {
    if (n == 0 || (n & (n - 1))) return NULL;
    yo_fft_plan *p = (yo_fft_plan *)calloc(1, sizeof(*p));
    p->n = n; p->dir = dir;
    p->rev = (uint32_t *)malloc(n * sizeof(uint32_t));
    p->tw = (cf32 *)malloc(n * sizeof(cf32));
    unsigned lg = 0; while (((size_t)1 << lg) < n) lg++;
    for (size_t i = 0; i < n; i++) {       /* plain bit reversal */
        size_t r = 0;
        for (unsigned b = 0; b < lg; b++) if (i & ((size_t)1 << b)) r |= (size_t)1 << (lg - 1 - b);
        p->rev[i] = (uint32_t)r;
    }
    const double s = dir ? 1.0 : -1.0;
    for (size_t m = 0; m < n; m++) {
        double a = s * 2.0 * M_PI * (double)m / (double)n;
        p->tw[m].re = (float)cos(a); p->tw[m].im = (float)sin(a);
    }
    return p;
}
void yo_fft_plan_destroy(yo_fft_plan *p) { if (p) { free(p->rev); free(p->tw); free(p); } }

/* out-of-place like Fft::run (copy, then in-place transform; fft/mod.rs:45-48) */
void yo_fft_run_f32(const yo_fft_plan *p, const cf32 *in, cf32 *out) {
    const size_t n = p->n;
    for (size_t i = 0; i < n; i++) out[p->rev[i]] = in[i];
    size_t len = 1;
    unsigned lg = 0; while (((size_t)1 << lg) < n) lg++;
    if (lg & 1) {                          /* one radix-2 pass */
        for (size_t i = 0; i < n; i += 2) {
            cf32 a = out[i], b = out[i + 1];
            out[i] = c_add(a, b); out[i + 1].re = a.re - b.re; out[i + 1].im = a.im - b.im;
        }
        len = 2;
    }
    const float sgn = p->dir ? 1.0f : -1.0f;
    /* radix-4 passes on bit-reversed data: a radix-4 DIT butterfly over bit-reversed order
       combines sub-blocks in order (0, 2, 1, 3) */
    for (; len < n; len *= 4) {
        const size_t step = n / (4 * len);
        for (size_t blk = 0; blk < n; blk += 4 * len) {
            for (size_t j = 0; j < len; j++) {
                cf32 w1 = p->tw[j * step], w2 = p->tw[2 * j * step], w3 = p->tw[3 * j * step];
                cf32 a = out[blk + j];
                cf32 b = c_mul(out[blk + j + 2 * len], w1);   /* bit-reversed: block 1 <-> 2 */
                cf32 c = c_mul(out[blk + j + len], w2);
                cf32 d = c_mul(out[blk + j + 3 * len], w3);
                cf32 t0 = c_add(a, c), t1 = { a.re - c.re, a.im - c.im };
                cf32 t2 = c_add(b, d), t3 = { b.re - d.re, b.im - d.im };
                /* multiply t3 by -j (forward) or +j (backward) */
                cf32 t3r = { -sgn * t3.im, sgn * t3.re };
                out[blk + j] = c_add(t0, t2);
                out[blk + j + len] = c_add(t1, t3r);
                out[blk + j + 2 * len].re = t0.re - t2.re; out[blk + j + 2 * len].im = t0.im - t2.im;
                out[blk + j + 3 * len].re = t1.re - t3r.re; out[blk + j + 3 * len].im = t1.im - t3r.im;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Kaiser-window FIR design, all in f32 like the reference.
 *   fir_design_kaiser                   src/filter/fir/design/kaiser.rs:16-51
 *   kaiser_beta_stopband_attenuation    kaiser.rs:62-72
 *   windows::kaiser                     src/math/windows.rs:76-90
 *   besseli0f/besselif/lnbesselif       src/math/bessel.rs:9-67
 *   lngammaf / gammaf (z>=0 branch)     src/math/gamma.rs:7-22 (+ gammaf = exp(lngammaf))
 *   sincf                               src/math/mod.rs:63-69
 * ---------------------------------------------------------------------------------- */
static const float PI_F = 3.14159265358979323846f;

float yo_lngammaf(float z) {
    if (z <= 0.0f) return NAN;               /* reference panics */
    if (z < 10.0f) return yo_lngammaf(z + 1.0f) - logf(z);
    float g = 0.5f * (logf(2.0f * PI_F) - logf(z));
    g += z * (logf(z + (1.0f / (12.0f * z - 0.1f / z))) - 1.0f);
    return g;
}
float yo_lnbesselif(float nu, float z) {
    if (z == 0.0f) return (nu == 0.0f) ? 0.0f : -INFINITY;
    if (nu == 0.5f) return 0.5f * logf(2.0f / (PI_F * z)) + logf(sinhf(z));
    if (z < 1e-3f * sqrtf(nu + 1.0f)) return -yo_lngammaf(nu + 1.0f) + nu * logf(0.5f * z);
    float t0 = nu * logf(0.5f * z);
    float y = 0.0f;
    for (int k = 0; k < 64; k++) {
        float t1 = 2.0f * (float)k * logf(0.5f * z);
        float t2 = yo_lngammaf((float)k + 1.0f);
        float t3 = yo_lngammaf(nu + (float)k + 1.0f);
        y += expf(t1 - t2 - t3);
    }
    return t0 + logf(y);
}
float yo_besseli0f(float z) {
    if (z == 0.0f) return 1.0f;
    /* besselif(0,z): low-signal branch (0.5 z)^0 / gammaf(1) = 1 / exp(lngammaf(1)) */
    if (z < 1e-3f) return powf(0.5f * z, 0.0f) / expf(yo_lngammaf(1.0f));
    return expf(yo_lnbesselif(0.0f, z));
}
float yo_sincf(float x) {
    if (fabsf(x) < 0.01f)
        return cosf(PI_F * x / 2.0f) * cosf(PI_F * x / 4.0f) * cosf(PI_F * x / 8.0f);
    return sinf(PI_F * x) / (PI_F * x);
}
float yo_kaiser_beta(float as_) {
    float a = fabsf(as_);
    if (a > 50.0f) return 0.1102f * (a - 8.7f);
    if (a > 21.0f) return 0.5842f * powf(a - 21.0f, 0.4f) + 0.07886f * (a - 21.0f);
    return 0.0f;
}
float yo_window_kaiser(size_t i, size_t wlen, float beta) {
    float t = (float)i - (float)(wlen - 1) / 2.0f;
    float r = 2.0f * t / (float)(wlen - 1);
    float a = yo_besseli0f(beta * sqrtf(1.0f - r * r));
    float b = yo_besseli0f(beta);
    return a / b;
}
int yo_fir_design_kaiser(size_t n, float fc, float as_, float mu, float *h) {
    if (mu <= -0.5f || mu > 0.5f) return YO_ECONFIG;
    if (fc <= 0.0f || fc > 0.5f) return YO_ECONFIG;
    if (n == 0) return YO_ECONFIG;
    if (as_ <= 0.0f) return YO_ECONFIG;
    float beta = yo_kaiser_beta(as_);
    for (size_t i = 0; i < n; i++) {
        float t = (float)i - ((float)n - 1.0f) / 2.0f + mu;
        float h1 = yo_sincf(2.0f * fc * t);
        float h2 = yo_window_kaiser(i, n, beta);
        h[i] = h1 * h2;
    }
    return YO_OK;
}

/* fir_design_notch, src/filter/fir/design/mod.rs:336-378 (f32): h = delta[m] - w(i) cos(2 pi f0 (i - m)) / scale */
int yo_fir_design_notch(size_t m, float f0, float as_, float *h) {
    if (m < 1 || m > 1000) return YO_ECONFIG;
    if (f0 < -0.5f || f0 > 0.5f) return YO_ECONFIG;
    if (as_ <= 0.0f) return YO_ECONFIG;
    const size_t n = 2 * m + 1;
    float beta = yo_kaiser_beta(as_);
    float scale = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float p = -cosf(2.0f * PI_F * f0 * ((float)i - (float)m));
        float w = yo_window_kaiser(i, n, beta);
        h[i] = p * w;
        scale += h[i] * p;
    }
    for (size_t i = 0; i < n; i++) h[i] /= scale;
    h[m] += 1.0f;
    return YO_OK;
}

/* ------------------------------------------------------------------------------------
 * taper windows, src/math/windows.rs:76-205 (f32).  type = WindowType discriminant:
 * 1 Hamming 2 Hann 3 BlackmanHarris 4 BlackmanHarris7 5 Kaiser 6 FlatTop 7 Triangular
 * 8 RcosTaper 9 Kbd.  Returns NaN where the reference returns Err.
 * ---------------------------------------------------------------------------------- */
float yo_window_fn(int type, size_t i, size_t wlen, float arg) {
    const float t = 2.0f * PI_F * (float)i / (float)(wlen - 1);
    switch (type) {
    case 1: return 0.53836f - 0.46164f * cosf((2.0f * PI_F * (float)i) / (float)(wlen - 1));
    case 2: return 0.5f - 0.5f * cosf((2.0f * PI_F * (float)i) / (float)(wlen - 1));
    case 3: return 0.35875f - 0.48829f * cosf(t) + 0.14128f * cosf(2.0f * t) - 0.01168f * cosf(3.0f * t);
    case 4: return 0.27105f - 0.43329f * cosf(t) + 0.21812f * cosf(2.0f * t) - 0.06592f * cosf(3.0f * t)
                 + 0.01081f * cosf(4.0f * t) - 0.00077f * cosf(5.0f * t) + 0.00001f * cosf(6.0f * t);
    case 5: return (i >= wlen || arg < 0.0f) ? NAN : yo_window_kaiser(i, wlen, arg);
    case 6: return 1.000f - 1.930f * cosf(t) + 1.290f * cosf(2.0f * t) - 0.388f * cosf(3.0f * t) + 0.028f * cosf(4.0f * t);
    case 7: {
        size_t n = (size_t)arg;
        if ((n != wlen - 1 && n != wlen && n != wlen + 1) || n == 0) return NAN;
        float v0 = (float)i - (float)(wlen - 1) / 2.0f, v1 = (float)n / 2.0f;
        return 1.0f - fabsf(v0 / v1);
    }
    case 8: {
        size_t tp = (size_t)arg;
        if (tp > wlen / 2) return NAN;
        if (i > wlen - tp - 1) i = wlen - i - 1;
        return (i < tp) ? 0.5f - 0.5f * cosf(PI_F * ((float)i + 0.5f) / (float)tp) : 1.0f;
    }
    case 9: {
        if (i >= wlen || wlen == 0 || (wlen % 2) != 0) return NAN;
        size_t m = wlen / 2;
        if (i >= m) return yo_window_fn(9, wlen - i - 1, wlen, arg);
        float w0 = 0.0f, w1 = 0.0f;
        for (size_t j = 0; j <= m; j++) {
            float w = yo_window_kaiser(j, m + 1, arg);
            w1 += w;
            if (j <= i) w0 += w;
        }
        return sqrtf(w0 / w1);
    }
    default: return NAN;
    }
}

/* ------------------------------------------------------------------------------------
 * Synthetic input generator: SplitMix64 used as a COUNTER-based
 * generator (draw u = mix(seed + (u+1)*gamma)), 24-bit uniforms like rand's gen::<f32>(),
 * Box-Muller as src/random/normal.rs:9-22 (real) and :29-44 scaled by 0.70710678 like
 * cawgn (:46-48) so re, im ~ N(0, 1/2).  u1 is drawn from (0,1] instead of rejecting 0.
 * ---------------------------------------------------------------------------------- */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void yo_gen_real(uint64_t seed, uint64_t first, size_t n, float *x) {
    for (size_t i = 0; i < n; i++) {
        uint64_t a = splitmix64_at(seed, 2 * (first + i)), b = splitmix64_at(seed, 2 * (first + i) + 1);
        float u1 = (float)((a >> 40) + 1) * (1.0f / 16777216.0f);
        float u2 = (float)(b >> 40) * (1.0f / 16777216.0f);
        x[i] = sqrtf(-2.0f * logf(u1)) * sinf(2.0f * PI_F * u2);
    }
}
void yo_gen_complex(uint64_t seed, uint64_t first, size_t n, cf32 *x) {
    for (size_t i = 0; i < n; i++) {
        uint64_t a = splitmix64_at(seed, 2 * (first + i)), b = splitmix64_at(seed, 2 * (first + i) + 1);
        float u1 = (float)((a >> 40) + 1) * (1.0f / 16777216.0f);
        float u2 = (float)(b >> 40) * (1.0f / 16777216.0f);
        float r = sqrtf(-2.0f * logf(u1)) * 0.70710678f, th = 2.0f * PI_F * u2;
        x[i].re = r * cosf(th); x[i].im = r * sinf(th);
    }
}

/* ------------------------------------------------------------------------------------
 * firpfbch / firpfbch2 analyzers -- NOT IN THE REFERENCE (multichannel/mod.rs is empty).
 * PARITY UNPINNED.  Composed from the reference's primitives with liquid-dsp's published
 * conventions (the library yagi rewrites; LIQUID_COMPAT.md:1765-1798 lists the absent
 * autotests):
 *   firpfbch analyzer, M channels, p taps per branch, prototype h[0..p*M):
 *     branch i taps h_sub[n] = h[i + n*M] stored reversed (exactly FirPfbFilter, firpfb.rs:45-52),
 *     one Window of length p per branch; push(x): w[idx].push(x); idx = (idx+M-1)%M, idx0 = M-1;
 *     run: X[M-1-i] = dot(branch_i, w[i]); y = forward DFT_M(X).  One frame = M pushes + run.
 *   firpfbch2 analyzer (2x oversampled), M even, m: h_len = 2*M*m, branch length 2m,
 *     one Window per branch; each step consumes M/2 inputs x[i] -> w[base-1-i], base = flag?M:M/2;
 *     X[b(i)] = dot(branch_i, w[b(i)]) with b(i) = (i + (flag ? M/2 : 0)) % M;
 *     y = IDFT_M(X)/M (unnormalised backward DFT then divide); flag ^= 1.
 * The DFT here is the f64 definition rounded to f32 at the end; branch dots are f32
 * sequential like dotprod.
 * ---------------------------------------------------------------------------------- */
typedef struct { size_t M, p, idx; float *filt; yo_window **w; } yo_firpfbch;

yo_firpfbch *yo_firpfbch_create(size_t M, size_t p, const float *h) {
    if (M == 0 || p == 0) return NULL;
    yo_firpfbch *q = (yo_firpfbch *)calloc(1, sizeof(*q));
    q->M = M; q->p = p; q->idx = M - 1;
    q->filt = (float *)malloc(M * p * sizeof(float));
    q->w = (yo_window **)malloc(M * sizeof(yo_window *));
    for (size_t i = 0; i < M; i++) {
        for (size_t n = 0; n < p; n++) q->filt[i * p + (p - n - 1)] = h[i + n * M];
        q->w[i] = yo_window_create(p, sizeof(cf32));
    }
    return q;
}
void yo_firpfbch_destroy(yo_firpfbch *q) {
    if (!q) return;
    for (size_t i = 0; i < q->M; i++) yo_window_destroy(q->w[i]);
    free(q->w); free(q->filt); free(q);
}
/* x: nframes*M inputs; y: nframes*M outputs [frame][channel] */
void yo_firpfbch_analyzer_execute(yo_firpfbch *q, const cf32 *x, size_t nframes, cf32 *y) {
    const size_t M = q->M, p = q->p;
    cf32 *X = (cf32 *)malloc(M * sizeof(cf32));
    cf64 *Y = (cf64 *)malloc(M * sizeof(cf64));
    for (size_t f = 0; f < nframes; f++) {
        for (size_t i = 0; i < M; i++) {
            yo_window_push(q->w[q->idx], &x[f * M + i]);
            q->idx = (q->idx + M - 1) % M;
        }
        for (size_t i = 0; i < M; i++) {
            const cf32 *r = (const cf32 *)yo_window_read(q->w[i]);
            yo_dotprod_rcc(q->filt + i * p, r, p, &X[M - 1 - i]);
        }
        yo_dft_f64(X, M, 0, Y);
        for (size_t k = 0; k < M; k++) { y[f * M + k].re = (float)Y[k].re; y[f * M + k].im = (float)Y[k].im; }
    }
    free(X); free(Y);
}

typedef struct { size_t M, m, flag; float *filt; yo_window **w; } yo_firpfbch2;

yo_firpfbch2 *yo_firpfbch2_create(size_t M, size_t m, const float *h) {
    if (M < 2 || (M & 1) || m < 1) return NULL;
    yo_firpfbch2 *q = (yo_firpfbch2 *)calloc(1, sizeof(*q));
    const size_t p = 2 * m;
    q->M = M; q->m = m; q->flag = 0;
    q->filt = (float *)malloc(M * p * sizeof(float));
    q->w = (yo_window **)malloc(M * sizeof(yo_window *));
    for (size_t i = 0; i < M; i++) {
        for (size_t n = 0; n < p; n++) q->filt[i * p + (p - n - 1)] = h[i + n * M];
        q->w[i] = yo_window_create(p, sizeof(cf32));
    }
    return q;
}
void yo_firpfbch2_destroy(yo_firpfbch2 *q) {
    if (!q) return;
    for (size_t i = 0; i < q->M; i++) yo_window_destroy(q->w[i]);
    free(q->w); free(q->filt); free(q);
}
/* x: nsteps*(M/2) inputs; y: nsteps*M outputs [step][channel] */
void yo_firpfbch2_analyzer_execute(yo_firpfbch2 *q, const cf32 *x, size_t nsteps, cf32 *y) {
    const size_t M = q->M, M2 = M / 2, p = 2 * q->m;
    cf32 *X = (cf32 *)malloc(M * sizeof(cf32));
    cf64 *Y = (cf64 *)malloc(M * sizeof(cf64));
    for (size_t s = 0; s < nsteps; s++) {
        size_t base = q->flag ? M : M2;
        for (size_t i = 0; i < M2; i++) yo_window_push(q->w[base - i - 1], &x[s * M2 + i]);
        size_t offset = q->flag ? M2 : 0;
        for (size_t i = 0; i < M; i++) {
            size_t b = (offset + i) % M;           /* window index and IDFT input slot */
            const cf32 *r = (const cf32 *)yo_window_read(q->w[b]);
            yo_dotprod_rcc(q->filt + i * p, r, p, &X[b]);
        }
        yo_dft_f64(X, M, 1, Y);
        for (size_t k = 0; k < M; k++) {
            y[s * M + k].re = (float)(Y[k].re / (double)M);
            y[s * M + k].im = (float)(Y[k].im / (double)M);
        }
        q->flag = 1 - q->flag;
    }
    free(X); free(Y);
}

/* ------------------------------------------------------------------------------------
 * FftFilt<T,Coeff>  --  src/filter/fftfilt.rs:22-142 (overlap-add, block n, FFT size 2n).
 * The reference's two transforms are rustfft calls (f32); here they are the f64 DFT
 * definition rounded to f32 after each transform (same hand-off points as the reference's
 * Complex32 buffers).  kind: 0 rrrf, 1 crcf, 2 cccf (x/y real for rrrf, h complex for cccf).
 * ---------------------------------------------------------------------------------- */
typedef struct { int kind; size_t h_len, n; cf32 *h_freq, *w, *tbuf, *fbuf; cf64 *tmp; cf32 scale; } yo_fftfilt;

static void yo_dft_f32io(const cf32 *in, size_t n, int dir, cf32 *out, cf64 *tmp) {
    yo_dft_f64(in, n, dir, tmp);
    for (size_t i = 0; i < n; i++) { out[i].re = (float)tmp[i].re; out[i].im = (float)tmp[i].im; }
}
void yo_fftfilt_set_scale(yo_fftfilt *q, float re, float im) {          /* :95-97 */
    float d = 2.0f * (float)q->n;
    q->scale.re = re / d; q->scale.im = im / d;
}
yo_fftfilt *yo_fftfilt_create(int kind, const float *h, size_t h_len, size_t n) {   /* :46-84 */
    if (h_len == 0 || n < h_len - 1 || n == 0) return NULL;
    yo_fftfilt *q = (yo_fftfilt *)calloc(1, sizeof(*q));
    q->kind = kind; q->h_len = h_len; q->n = n;
    q->h_freq = (cf32 *)calloc(2 * n, sizeof(cf32)); q->w = (cf32 *)calloc(n, sizeof(cf32));
    q->tbuf = (cf32 *)calloc(2 * n, sizeof(cf32)); q->fbuf = (cf32 *)calloc(2 * n, sizeof(cf32));
    q->tmp = (cf64 *)calloc(2 * n, sizeof(cf64));
    for (size_t i = 0; i < 2 * n; i++) {
        if (i < h_len) { q->tbuf[i].re = kind == 2 ? h[2 * i] : h[i]; q->tbuf[i].im = kind == 2 ? h[2 * i + 1] : 0.0f; }
        else q->tbuf[i] = CZERO;
    }
    yo_dft_f32io(q->tbuf, 2 * n, 0, q->h_freq, q->tmp);
    yo_fftfilt_set_scale(q, 1.0f, 0.0f);
    return q;
}
void yo_fftfilt_destroy(yo_fftfilt *q) {
    if (q) { free(q->h_freq); free(q->w); free(q->tbuf); free(q->fbuf); free(q->tmp); free(q); }
}
void yo_fftfilt_reset(yo_fftfilt *q) { memset(q->w, 0, q->n * sizeof(cf32)); }      /* :86-88 */
void yo_fftfilt_execute(yo_fftfilt *q, const float *x, float *y) {                  /* :103-138 */
    const size_t n = q->n;
    for (size_t i = 0; i < n; i++) {
        if (q->kind == 0) { q->tbuf[i].re = x[i]; q->tbuf[i].im = 0.0f; }
        else { q->tbuf[i].re = x[2 * i]; q->tbuf[i].im = x[2 * i + 1]; }
    }
    for (size_t i = n; i < 2 * n; i++) q->tbuf[i] = CZERO;
    yo_dft_f32io(q->tbuf, 2 * n, 0, q->fbuf, q->tmp);
    for (size_t i = 0; i < 2 * n; i++) q->fbuf[i] = c_mul(q->fbuf[i], q->h_freq[i]);
    yo_dft_f32io(q->fbuf, 2 * n, 1, q->tbuf, q->tmp);
    for (size_t i = 0; i < n; i++) {
        cf32 v = c_mul(c_add(q->tbuf[i], q->w[i]), q->scale);
        if (q->kind == 0) y[i] = v.re; else { y[2 * i] = v.re; y[2 * i + 1] = v.im; }
    }
    memcpy(q->w, q->tbuf + n, n * sizeof(cf32));
}

/* the headline stream: firfilt_crcf execute_block, output cut into
 * consecutive nfft-sample frames, forward FFT of each (f32 radix-4 plan above).
 * Used by tests (small) and as the timed cpu_baseline of bench.py. */
void yo_stream_fir_fft(yo_firfilt_crcf *q, const yo_fft_plan *p, const cf32 *x, size_t nframes,
                       cf32 *scratch, cf32 *spectra) {
    const size_t n = p->n;
    for (size_t f = 0; f < nframes; f++) {
        yo_firfilt_crcf_execute_block(q, x + f * n, n, scratch, n);
        yo_fft_run_f32(p, scratch, spectra + f * n);
    }
}

/* ------------------------------------------------------------------------------------
 * Resamp2<T,Coeff>  --  src/filter/resampler/resamp2.rs:26-174
 * MsResamp2<T,Coeff> -- src/filter/resampler/msresamp2.rs:8-198
 *
 * new() (resamp2.rs:44-88) designs its half-band prototype with fir_design_pm_halfband_stopband_attenuation
 * (Parks-McClellan design code: out of the hot path's scope).  The restatement therefore takes the designed
 * prototype hf[4m+1] as an argument and follows the reference from there: modulation (for_halfband :12-22), the
 * 2m taps h1[i] = h[h_len - 2i - 2] (:66-70), two Window<T> of 2m samples, toggle.
 * Arithmetic is spelled as the reference spells it: 0.5 * (yi + yq) * scale etc.
 * ---------------------------------------------------------------------------------- */
static const float YO_PI_F = 3.14159265358979323846f;   /* std::f32::consts::PI */
static inline float r_sub(float a, float b) { return a - b; }
static inline cf32 c_sub(cf32 a, cf32 b) { cf32 r = { a.re - b.re, a.im - b.im }; return r; }
static inline float r_half(float a) { return 0.5f * a; }
static inline cf32 c_half(cf32 a) { cf32 r = { 0.5f * a.re, 0.5f * a.im }; return r; }   /* T::from(0.5) * a, im(0.5) = 0: */
/* Complex(0.5,0) * a = (0.5 a.re - 0 a.im, 0.5 a.im + 0 a.re): equal bit for bit to the component form for finite a */
static inline float halfband_r(float hf, float t, float f0) { return 2.0f * hf * cosf(2.0f * YO_PI_F * t * f0); }
static inline cf32 halfband_c(float hf, float t, float f0) {
    float g = 2.0f * hf, a = 2.0f * YO_PI_F * t * f0;
    cf32 r = { g * cosf(a), g * sinf(a) };
    return r;
}
static inline float dot_rr(const float *h, const float *r, size_t n) { return yo_dotprod_rrrf(h, r, n); }
static inline cf32 dot_rc(const float *h, const cf32 *r, size_t n) { cf32 y; yo_dotprod_rcc(h, r, n, &y); return y; }
static inline cf32 dot_cc(const cf32 *h, const cf32 *r, size_t n) { cf32 y; yo_dotprod_ccc(h, r, n, &y); return y; }

#define DEFINE_RESAMP2(SUF, T, C, MUL_TC, ADD_T, SUB_T, HALF_T, ONE_C, DOT, HALFBAND)              \
typedef struct { size_t m; C *h1; yo_window *w0, *w1; C scale; int toggle; } yo_resamp2_##SUF;    \
yo_resamp2_##SUF *yo_resamp2_##SUF##_create(const float *hf, size_t m, float f0) { /* :44-88 */   \
    if (m < 2) return NULL;                                                                       \
    if (f0 < -0.5f || f0 > 0.5f) return NULL;                                                     \
    size_t h_len = 4 * m + 1;                                                                     \
    C *h = (C *)malloc(h_len * sizeof(C));                                                        \
    for (size_t i = 0; i < h_len; i++) {                                                          \
        float t = (float)i - (float)(h_len - 1) / 2.0f;                                           \
        h[i] = HALFBAND(hf[i], t, f0);                                                            \
    }                                                                                             \
    yo_resamp2_##SUF *q = (yo_resamp2_##SUF *)calloc(1, sizeof(*q));                              \
    q->m = m; q->h1 = (C *)malloc(2 * m * sizeof(C));                                             \
    for (size_t i = 0; i < 2 * m; i++) q->h1[i] = h[h_len - 2 * i - 2];                           \
    free(h);                                                                                      \
    q->w0 = yo_window_create(2 * m, sizeof(T)); q->w1 = yo_window_create(2 * m, sizeof(T));       \
    q->scale = ONE_C; q->toggle = 0;                                                              \
    return q;                                                                                     \
}                                                                                                 \
void yo_resamp2_##SUF##_destroy(yo_resamp2_##SUF *q) {                                            \
    if (q) { free(q->h1); yo_window_destroy(q->w0); yo_window_destroy(q->w1); free(q); }          \
}                                                                                                 \
yo_resamp2_##SUF *yo_resamp2_##SUF##_clone(const yo_resamp2_##SUF *s) {      /* derive(Clone) :25 */ \
    yo_resamp2_##SUF *q = (yo_resamp2_##SUF *)calloc(1, sizeof(*q));                              \
    q->m = s->m; q->h1 = (C *)malloc(2 * s->m * sizeof(C)); memcpy(q->h1, s->h1, 2 * s->m * sizeof(C)); \
    q->w0 = yo_window_clone(s->w0); q->w1 = yo_window_clone(s->w1); q->scale = s->scale; q->toggle = s->toggle; \
    return q;                                                                                     \
}                                                                                                 \
void yo_resamp2_##SUF##_reset(yo_resamp2_##SUF *q) {                          /* :90-94 */        \
    yo_window_reset(q->w0); yo_window_reset(q->w1); q->toggle = 0;                                \
}                                                                                                 \
void yo_resamp2_##SUF##_set_scale(yo_resamp2_##SUF *q, C s) { q->scale = s; } /* :96-98 */        \
void yo_resamp2_##SUF##_filter_execute(yo_resamp2_##SUF *q, T x, T *y0, T *y1) { /* :108-130 */   \
    T yi, yq;                                                                                     \
    if (!q->toggle) {                                                                             \
        yo_window_push(q->w0, &x); yo_window_index(q->w0, q->m - 1, &yi);                         \
        yq = DOT(q->h1, (const T *)yo_window_read(q->w1), 2 * q->m);                              \
    } else {                                                                                      \
        yo_window_push(q->w1, &x); yo_window_index(q->w1, q->m - 1, &yi);                         \
        yq = DOT(q->h1, (const T *)yo_window_read(q->w0), 2 * q->m);                              \
    }                                                                                             \
    q->toggle = !q->toggle;                                                                       \
    *y0 = MUL_TC(HALF_T(ADD_T(yi, yq)), q->scale);                                                \
    *y1 = MUL_TC(HALF_T(SUB_T(yi, yq)), q->scale);                                                \
}                                                                                                 \
void yo_resamp2_##SUF##_analyzer_execute(yo_resamp2_##SUF *q, const T *x, T *y) { /* :132-143 */  \
    T a = HALF_T(x[0]), b = HALF_T(x[1]), y0;                                                     \
    yo_window_push(q->w1, &a);                                                                    \
    T y1 = DOT(q->h1, (const T *)yo_window_read(q->w1), 2 * q->m);                                \
    yo_window_push(q->w0, &b); yo_window_index(q->w0, q->m - 1, &y0);                             \
    y[0] = MUL_TC(ADD_T(y1, y0), q->scale);                                                       \
    y[1] = MUL_TC(SUB_T(y1, y0), q->scale);                                                       \
}                                                                                                 \
void yo_resamp2_##SUF##_synthesizer_execute(yo_resamp2_##SUF *q, const T *x, T *y) { /* :145-157 */ \
    T x0 = ADD_T(x[0], x[1]), x1 = SUB_T(x[0], x[1]), d;                                          \
    yo_window_push(q->w0, &x0); yo_window_index(q->w0, q->m - 1, &d);                             \
    y[0] = MUL_TC(d, q->scale);                                                                   \
    yo_window_push(q->w1, &x1);                                                                   \
    y[1] = MUL_TC(DOT(q->h1, (const T *)yo_window_read(q->w1), 2 * q->m), q->scale);              \
}                                                                                                 \
T yo_resamp2_##SUF##_decim_execute(yo_resamp2_##SUF *q, const T *x) {        /* :159-169 */       \
    T y0;                                                                                         \
    yo_window_push(q->w1, &x[0]);                                                                 \
    T y1 = DOT(q->h1, (const T *)yo_window_read(q->w1), 2 * q->m);                                \
    yo_window_push(q->w0, &x[1]); yo_window_index(q->w0, q->m - 1, &y0);                          \
    return MUL_TC(ADD_T(y0, y1), q->scale);                                                       \
}                                                                                                 \
void yo_resamp2_##SUF##_interp_execute(yo_resamp2_##SUF *q, T x, T *y) {     /* :171-180 */       \
    T d;                                                                                          \
    yo_window_push(q->w0, &x); yo_window_index(q->w0, q->m - 1, &d);                              \
    y[0] = MUL_TC(d, q->scale);                                                                   \
    yo_window_push(q->w1, &x);                                                                    \
    y[1] = MUL_TC(DOT(q->h1, (const T *)yo_window_read(q->w1), 2 * q->m), q->scale);              \
}                                                                                                 \
/* block forms: the per-call functions above, n times (mode 0 filter: n in -> 2n out as (y0,y1) pairs;           \
 * 1 analyzer / 2 synthesizer: n pairs -> n pairs; 3 decim: 2n -> n; 4 interp: n -> 2n) */                       \
void yo_resamp2_##SUF##_execute_block(yo_resamp2_##SUF *q, int mode, const T *x, size_t n, T *y) {               \
    for (size_t i = 0; i < n; i++) {                                                              \
        switch (mode) {                                                                           \
        case 0: yo_resamp2_##SUF##_filter_execute(q, x[i], &y[2 * i], &y[2 * i + 1]); break;      \
        case 1: yo_resamp2_##SUF##_analyzer_execute(q, x + 2 * i, y + 2 * i); break;              \
        case 2: yo_resamp2_##SUF##_synthesizer_execute(q, x + 2 * i, y + 2 * i); break;           \
        case 3: y[i] = yo_resamp2_##SUF##_decim_execute(q, x + 2 * i); break;                     \
        default: yo_resamp2_##SUF##_interp_execute(q, x[i], y + 2 * i); break;                    \
        }                                                                                         \
    }                                                                                             \
}                                                                                                 \
                                                                                                  \
/* ---- MsResamp2: msresamp2.rs:8-198; stage prototypes hf_s[4 m_s + 1] supplied (design out of scope) ---- */  \
typedef struct { int interp; size_t num_stages, rate; C zeta; T *b0, *b1; yo_resamp2_##SUF **st; } yo_msresamp2_##SUF; \
yo_msresamp2_##SUF *yo_msresamp2_##SUF##_create(int interp, size_t num_stages, const size_t *m_stage,            \
                                                 const float *hf_all, C zeta) {   /* :38-93 */    \
    if (num_stages > 16) return NULL;                                                             \
    yo_msresamp2_##SUF *q = (yo_msresamp2_##SUF *)calloc(1, sizeof(*q));                          \
    q->interp = interp; q->num_stages = num_stages; q->rate = (size_t)1 << num_stages; q->zeta = zeta; \
    q->b0 = (T *)calloc(q->rate, sizeof(T)); q->b1 = (T *)calloc(q->rate, sizeof(T));             \
    q->st = (yo_resamp2_##SUF **)calloc(num_stages ? num_stages : 1, sizeof(*q->st));             \
    for (size_t i = 0; i < num_stages; i++) {                                                     \
        q->st[i] = yo_resamp2_##SUF##_create(hf_all, m_stage[i], 0.0f);      /* f0_stage = 0 (:44-46) */ \
        hf_all += 4 * m_stage[i] + 1;                                                             \
    }                                                                                             \
    return q;                                                                                     \
}                                                                                                 \
void yo_msresamp2_##SUF##_destroy(yo_msresamp2_##SUF *q) {                                        \
    if (!q) return;                                                                               \
    for (size_t i = 0; i < q->num_stages; i++) yo_resamp2_##SUF##_destroy(q->st[i]);              \
    free(q->st); free(q->b0); free(q->b1); free(q);                                               \
}                                                                                                 \
void yo_msresamp2_##SUF##_interp_execute(yo_msresamp2_##SUF *q, T x, T *y) {  /* :154-175 */      \
    T *b0 = q->b0, *b1 = q->b1;                                                                   \
    b0[0] = x;                                                                                    \
    for (size_t s = 0; s < q->num_stages; s++) {                                                  \
        size_t k = (size_t)1 << s;                                                                \
        for (size_t i = 0; i < k; i++) yo_resamp2_##SUF##_interp_execute(q->st[s], b0[i], &b1[2 * i]); \
        T *t = b0; b0 = b1; b1 = t;                                                               \
    }                                                                                             \
    memcpy(y, b0, q->rate * sizeof(T));                                                           \
}                                                                                                 \
T yo_msresamp2_##SUF##_decim_execute(yo_msresamp2_##SUF *q, const T *x) {     /* :177-197 */      \
    T *b0 = q->b0, *b1 = q->b1;                                                                   \
    memcpy(b0, x, q->rate * sizeof(T));                                                           \
    for (size_t s = 0; s < q->num_stages; s++) {                                                  \
        size_t k = (size_t)1 << (q->num_stages - s - 1), g = q->num_stages - s - 1;               \
        for (size_t i = 0; i < k; i++) b1[i] = yo_resamp2_##SUF##_decim_execute(q->st[g], &b0[2 * i]); \
        T *t = b0; b0 = b1; b1 = t;                                                               \
    }                                                                                             \
    return MUL_TC(b0[0], q->zeta);                                                                \
}                                                                                                 \
/* n execute() calls (:137-152): interp n -> n*rate, decim n*rate -> n; num_stages = 0 copies */  \
void yo_msresamp2_##SUF##_execute_block(yo_msresamp2_##SUF *q, const T *x, size_t n, T *y) {      \
    for (size_t i = 0; i < n; i++) {                                                              \
        if (q->num_stages == 0) y[i] = x[i];                                                      \
        else if (q->interp) yo_msresamp2_##SUF##_interp_execute(q, x[i], y + i * q->rate);        \
        else y[i] = yo_msresamp2_##SUF##_decim_execute(q, x + i * q->rate);                       \
    }                                                                                             \
}

DEFINE_RESAMP2(rrrf, float, float, r_mul, r_add, r_sub, r_half, 1.0f, dot_rr, halfband_r)
DEFINE_RESAMP2(crcf, cf32, float, c_mulr, c_add, c_sub, c_half, 1.0f, dot_rc, halfband_r)
DEFINE_RESAMP2(cccf, cf32, cf32, c_mul, c_add, c_sub, c_half, CONE, dot_cc, halfband_c)

/* estimate_req_filter_len (design/mod.rs:138-152, Kaiser's formula :228-238) and the per-stage semi-lengths of
 * MsResamp2::new (msresamp2.rs:70-88); returns 0 on a configuration error */
size_t yo_estimate_req_filter_len(float df, float as_) {
    if (df <= 0.0f || df > 0.5f || as_ <= 0.0f) return 0;
    float h_len = (as_ - 7.95f) / (14.26f * df);
    return (size_t)h_len;                    /* `n as usize`: saturating truncation toward zero */
}
int yo_msresamp2_stage_lengths(size_t num_stages, float fc, float as_, size_t *m_stage) {
    if (num_stages > 16 || fc <= 0.0f || fc >= 0.5f) return YO_ECONFIG;
    float a = as_ + 5.0f;
    for (size_t i = 0; i < num_stages; i++) {
        fc = (i == 1) ? (0.5f - fc) / 2.0f : 0.5f * fc;
        float ft = 2.0f * (0.25f - fc);
        if (ft <= 0.0f || ft > 0.5f || a <= 0.0f) return YO_ECONFIG;
        float hl = (a - 7.95f) / (14.26f * ft);
        size_t h_len = hl <= 0.0f ? 0 : (size_t)hl;
        size_t m = (size_t)ceilf(((float)h_len - 1.0f) / 4.0f);
        m_stage[i] = m < 3 ? 3 : m;
    }
    return YO_OK;
}
