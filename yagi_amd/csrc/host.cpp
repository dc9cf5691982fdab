// host.cpp -- host-only pieces of libyagi_hip.so: error reporting and the design-time scalar
// maths the constructors need (Kaiser-windowed sinc taps).  Design code runs once per object
// on the CPU in the reference as well; it is not on the hot path.
#include <cmath>

#include "common.hpp"

namespace yagi {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int fail(int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

// translate the exception in flight (called from a catch (...) block) into a status + message
int api_exception() noexcept {
    const char *what = "unexpected C++ exception";
    std::string msg;
    try {
        throw;
    } catch (const std::bad_alloc &) {
        what = "out of host memory";
    } catch (const std::exception &e) {
        try { msg = std::string("unexpected C++ exception: ") + e.what(); what = msg.c_str(); } catch (...) {}
    } catch (...) {
    }
    try { g_last_error = what; } catch (...) {}
    return YAGI_ERR_INTERNAL;
}

// ---- Kaiser design, single precision like the reference -----------------------------------
// ln Gamma(z): recursion below 10, Stirling-type series above (math/gamma.rs:7-22)
static float ln_gamma(float z) {
    // below 10 step up with lnGamma(z) = lnGamma(z+1) - ln z; the subtractions are applied
    // largest argument first, which is the order the reference's recursion produces
    float logs[16];
    int nlog = 0;
    while (z < 10.0f && nlog < 16) {
        logs[nlog++] = std::log(z);
        z += 1.0f;
    }
    const float two_pi = 2.0f * 3.14159265358979323846f;
    float g = 0.5f * (std::log(two_pi) - std::log(z));
    g += z * (std::log(z + 1.0f / (12.0f * z - 0.1f / z)) - 1.0f);
    for (int i = nlog - 1; i >= 0; --i) g -= logs[i];
    return g;
}

// I0(z) through the log-domain series the reference uses (math/bessel.rs:9-67)
static float bessel_i0(float z) {
    if (z == 0.0f) return 1.0f;
    if (z < 1e-3f) return 1.0f / std::exp(ln_gamma(1.0f));
    const float lhz = std::log(0.5f * z);
    float acc = 0.0f;
    for (int k = 0; k < 64; ++k) {
        const float kf = static_cast<float>(k);
        const float lg = ln_gamma(kf + 1.0f);
        acc += std::exp(2.0f * kf * lhz - lg - lg);
    }
    return std::exp(0.0f + std::log(acc));
}

static float sinc(float x) {     // math/mod.rs:63-69
    const float pi = 3.14159265358979323846f;
    if (std::fabs(x) < 0.01f)
        return std::cos(pi * x / 2.0f) * std::cos(pi * x / 4.0f) * std::cos(pi * x / 8.0f);
    return std::sin(pi * x) / (pi * x);
}

static float kaiser_beta(float as_) {     // kaiser.rs:62-72
    const float a = std::fabs(as_);
    if (a > 50.0f) return 0.1102f * (a - 8.7f);
    if (a > 21.0f) return 0.5842f * std::pow(a - 21.0f, 0.4f) + 0.07886f * (a - 21.0f);
    return 0.0f;
}

// fir_design_kaiser (kaiser.rs:16-51) with windows::kaiser (math/windows.rs:76-90)
int design_kaiser(size_t n, float fc, float as_, float mu, std::vector<float> &h) {
    if (mu <= -0.5f || mu > 0.5f)
        return fail(YAGI_ERR_CONFIG, "fractional sample offset (%g) out of range (-0.5, 0.5)", mu);
    if (fc <= 0.0f || fc > 0.5f)
        return fail(YAGI_ERR_CONFIG, "cutoff frequency (%g) out of range (0, 0.5)", fc);
    if (n == 0) return fail(YAGI_ERR_CONFIG, "filter length must be greater than zero");
    if (as_ <= 0.0f) return fail(YAGI_ERR_CONFIG, "stop-band attenuation must be greater than zero");
    const float beta = kaiser_beta(as_);
    const float i0_beta = bessel_i0(beta);
    h.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const float t = static_cast<float>(i) - (static_cast<float>(n) - 1.0f) / 2.0f + mu;
        const float proto = sinc(2.0f * fc * t);
        const float tw = static_cast<float>(i) - static_cast<float>(n - 1) / 2.0f;
        const float r = 2.0f * tw / static_cast<float>(n - 1);
        const float win = bessel_i0(beta * std::sqrt(1.0f - r * r)) / i0_beta;
        h[i] = proto * win;
    }
    return YAGI_OK;
}

// fir_design_notch (design/mod.rs:336-378): 1 - (Kaiser-windowed tone at f0, normalised to unit gain at f0)
int design_notch(size_t m, float f0, float as_, std::vector<float> &h) {
    if (m < 1 || m > 1000) return fail(YAGI_ERR_CONFIG, "filter semi-length (%zu) out of range [1,1000]", m);
    if (f0 < -0.5f || f0 > 0.5f) return fail(YAGI_ERR_CONFIG, "notch frequency (%g) out of range [-0.5,0.5]", (double)f0);
    if (as_ <= 0.0f) return fail(YAGI_ERR_CONFIG, "stop-band attenuation must be greater than zero");
    const size_t n = 2 * m + 1;
    const float beta = kaiser_beta(as_);
    const float i0_beta = bessel_i0(beta);
    h.assign(n, 0.0f);
    float scale = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        const float p = -std::cos(2.0f * 3.14159265358979323846f * f0 * (static_cast<float>(i) - static_cast<float>(m)));
        const float tw = static_cast<float>(i) - static_cast<float>(n - 1) / 2.0f;
        const float r = 2.0f * tw / static_cast<float>(n - 1);
        const float w = bessel_i0(beta * std::sqrt(1.0f - r * r)) / i0_beta;
        h[i] = p * w;
        scale += h[i] * p;
    }
    for (size_t i = 0; i < n; ++i) h[i] /= scale;
    h[m] += 1.0f;
    return YAGI_OK;
}

// ---- taper windows for Spgram (math/windows.rs:76-205), single precision like the reference -------
// type: 1 Hamming, 2 Hann, 3 BlackmanHarris, 4 BlackmanHarris7, 5 Kaiser, 6 FlatTop, 7 Triangular,
//       8 RcosTaper, 9 Kbd (the reference's WindowType discriminants)
static float kaiser_window(size_t i, size_t wlen, float beta) {
    const float t = static_cast<float>(i) - static_cast<float>(wlen - 1) / 2.0f;
    const float r = 2.0f * t / static_cast<float>(wlen - 1);
    return bessel_i0(beta * std::sqrt(1.0f - r * r)) / bessel_i0(beta);
}

int window_value(int type, size_t i, size_t wlen, float arg, float *out) {
    const float pi = 3.14159265358979323846f;
    const float t = 2.0f * pi * static_cast<float>(i) / static_cast<float>(wlen - 1);
    switch (type) {
        case 1: *out = 0.53836f - 0.46164f * std::cos((2.0f * pi * static_cast<float>(i)) / static_cast<float>(wlen - 1)); return YAGI_OK;
        case 2: *out = 0.5f - 0.5f * std::cos((2.0f * pi * static_cast<float>(i)) / static_cast<float>(wlen - 1)); return YAGI_OK;
        case 3: *out = 0.35875f - 0.48829f * std::cos(t) + 0.14128f * std::cos(2.0f * t) - 0.01168f * std::cos(3.0f * t); return YAGI_OK;
        case 4:
            *out = 0.27105f - 0.43329f * std::cos(t) + 0.21812f * std::cos(2.0f * t) - 0.06592f * std::cos(3.0f * t) +
                   0.01081f * std::cos(4.0f * t) - 0.00077f * std::cos(5.0f * t) + 0.00001f * std::cos(6.0f * t);
            return YAGI_OK;
        case 5:
            if (arg < 0.0f) return fail(YAGI_ERR_VALUE, "Kaiser window: beta must be greater than or equal to zero");
            *out = kaiser_window(i, wlen, arg);
            return YAGI_OK;
        case 6: *out = 1.000f - 1.930f * std::cos(t) + 1.290f * std::cos(2.0f * t) - 0.388f * std::cos(3.0f * t) + 0.028f * std::cos(4.0f * t); return YAGI_OK;
        case 7: {
            const size_t n = static_cast<size_t>(arg);
            if (n + 1 != wlen && n != wlen && n != wlen + 1)
                return fail(YAGI_ERR_VALUE, "Triangular window: sub-length must be in wlen+{-1,0,1}");
            if (n == 0) return fail(YAGI_ERR_VALUE, "Triangular window: sub-length must be greater than zero");
            const float v0 = static_cast<float>(i) - static_cast<float>(wlen - 1) / 2.0f;
            *out = 1.0f - std::fabs(v0 / (static_cast<float>(n) / 2.0f));
            return YAGI_OK;
        }
        case 8: {
            const size_t tp = static_cast<size_t>(arg);
            if (tp > wlen / 2) return fail(YAGI_ERR_VALUE, "Raised-cosine taper window: taper length cannot exceed half window length");
            size_t ii = i;
            if (ii > wlen - tp - 1) ii = wlen - ii - 1;
            *out = (ii < tp) ? 0.5f - 0.5f * std::cos(pi * (static_cast<float>(ii) + 0.5f) / static_cast<float>(tp)) : 1.0f;
            return YAGI_OK;
        }
        case 9: {
            if (wlen % 2 != 0) return fail(YAGI_ERR_VALUE, "KBD window: window length must be even");
            const size_t m = wlen / 2;
            const size_t ii = (i >= m) ? wlen - i - 1 : i;
            float w0 = 0.0f, w1 = 0.0f;
            for (size_t j = 0; j <= m; ++j) {
                const float wv = kaiser_window(j, m + 1, arg);
                w1 += wv;
                if (j <= ii) w0 += wv;
            }
            *out = std::sqrt(w0 / w1);
            return YAGI_OK;
        }
        default: return fail(YAGI_ERR_CONFIG, "unknown window type");
    }
}

// ---------------------------------------------------------------------------------------------
// Per-sample arithmetic of the FIR objects on the host mirror of their window (capi.hip DevWindow): what
// FirFilter::execute (firfilt.rs:241-246), FirPfbFilter::execute (firpfb.rs:277-286) and
// FirDecimationFilter::execute (firdecim.rs:179-191) compute, in the reference's order: products unfused, every
// sum left to right from zero (dotprod/mod.rs:19-73: `iter().zip().map(|(a, b)| a * b).sum()`), the scale applied to
// the finished sum.  This is product code (tens of nanoseconds per call), not the test oracle.
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
namespace {
inline float h_mul(float a, float b) { return a * b; }
inline cf32 h_mul(cf32 a, float b) { return cf32{a.re * b, a.im * b}; }
inline cf32 h_mul(cf32 a, cf32 b) { return cf32{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline float h_add(float a, float b) { return a + b; }
inline cf32 h_add(cf32 a, cf32 b) { return cf32{a.re + b.re, a.im + b.im}; }
}  // namespace

// FirFilter: the state is a VecDeque with the NEWEST sample first (push = rotate_right(1) + w[0] = x, :220-223), summed
// against h[0..L) as two slices (dotprod/mod.rs:75-121): the ring wraps `split = L - head` samples from its start.
// w = the window oldest first (w[L-1] = newest), head = the VecDeque's physical head.
template <class T, class C>
T host_fir_ring_dot(const T *w, size_t L, size_t head, const C *h, C scale) {
    const size_t split = L - head;
    T l{}, r{};
    const T *newest = w + (L - 1);
    for (size_t k = 0; k < split; ++k) l = h_add(l, h_mul(newest[-(ptrdiff_t)k], h[k]));
    for (size_t k = split; k < L; ++k) r = h_add(r, h_mul(newest[-(ptrdiff_t)k], h[k]));
    return h_mul(h_add(l, r), scale);
}
// Window-based objects: the contiguous window oldest first against the taps reversed (firdecim.rs:47, firpfb.rs:45-52);
// h = the taps in natural order (h[0] meets the newest sample)
template <class T, class C>
T host_fir_window_dot(const T *w, size_t L, const C *h, C scale) {
    T s{};
    for (size_t k = 0; k < L; ++k) s = h_add(s, h_mul(w[k], h[L - 1 - k]));
    return h_mul(s, scale);
}
template float host_fir_ring_dot<float, float>(const float *, size_t, size_t, const float *, float);
template cf32 host_fir_ring_dot<cf32, float>(const cf32 *, size_t, size_t, const float *, float);
template cf32 host_fir_ring_dot<cf32, cf32>(const cf32 *, size_t, size_t, const cf32 *, cf32);
template float host_fir_window_dot<float, float>(const float *, size_t, const float *, float);
template cf32 host_fir_window_dot<cf32, float>(const cf32 *, size_t, const float *, float);
template cf32 host_fir_window_dot<cf32, cf32>(const cf32 *, size_t, const cf32 *, cf32);

}  // namespace yagi

extern "C" {

const char *yagi_hip_last_error(void) { return yagi::g_last_error.c_str(); }
const char *yagi_hip_version(void) { return "yagi_hip 0.1.0 (gfx950)"; }

int yagi_hip_fir_design_kaiser(size_t n, float fc, float as_, float mu, float *h) {
    std::vector<float> v;
    YG_TRY(yagi::design_kaiser(n, fc, as_, mu, v));
    if (!h) return yagi::fail(YAGI_ERR_CONFIG, "null output pointer");
    std::memcpy(h, v.data(), n * sizeof(float));
    return YAGI_OK;
}
}
