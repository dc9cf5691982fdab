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
