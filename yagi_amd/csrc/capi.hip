// capi.hip -- the extern "C" surface of libyagi_hip.so (include/yagi_hip.h): stateful objects that
// mirror yagi's FirFilter / FirDecimationFilter / FirPfbFilter / Fft (+ the channelizers and the
// fused FIR->FFT stream) with all arithmetic on the device.
//
// State model.  The reference keeps a Window<T>/VecDeque<T> of the last L samples inside each
// object (firfilt.rs:13, firdecim.rs:15, firpfb.rs:12).  Here that window lives in HBM as L
// samples, oldest first (a ring of three buffers so an update never races the kernel reading it),
// with a host mirror for the per-sample calls: push() / execute() / execute_one() run on the host
// mirror with the reference's own sequential sums (host.cpp: tens of nanoseconds per call, no
// launch), block calls run the FIR kernels on the device window; the two copies are synchronised
// lazily when the caller switches between the two kinds of call (DevWindow).  clone() copies the
// state; reset() zeroes it.
#include <cmath>
#include <cstdlib>
#include <memory>

#include <algorithm>

#include "kernels.hpp"

namespace yagi {

int design_kaiser(size_t n, float fc, float as_, float mu, std::vector<float> &h);   // host.cpp
int window_value(int type, size_t i, size_t wlen, float arg, float *out);            // host.cpp
int design_notch(size_t m, float f0, float as_, std::vector<float> &h);                // host.cpp

static int require_device() {
    static int ok = -1;
    if (ok < 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        ok = (e == hipSuccess && n > 0) ? 1 : 0;
    }
    if (!ok) return fail(YAGI_ERR_DEVICE, "no HIP device available: libyagi_hip has no CPU fallback");
    return YAGI_OK;
}

template <class V> static V one_of();
template <> float one_of<float>() { return 1.0f; }
template <> cf32 one_of<cf32>() { return cf32{1.0f, 0.0f}; }
static cf32 to_c(float v, cf32 *) { return cf32{v, 0.0f}; }
static float to_c(float v, float *) { return v; }

static int upload(void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return YAGI_OK;
    YG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    YG_HIP(hipStreamSynchronize(st));
    return YAGI_OK;
}
static int download(void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return YAGI_OK;
    YG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    YG_HIP(hipStreamSynchronize(st));
    return YAGI_OK;
}

// Pipelined block calls (yagi_hip_firfft_crcf_set_pipeline, yagi_hip_firfilt_*_set_pipeline).  Consecutive blocks of the stream depend on each other
// only through the L-sample filter window, and that window is INPUT data (the previous block's last L samples), which
// the pipelined contract keeps intact until the join: block b + 1 reads it straight from the previous call's x, so it
// needs nothing block b computes.  The block kernels alternate between two streams (lanes); call b does
//     caller's stream: record `in`           lane b % 2: wait(in), block kernel b, record done[b % 2]
// so block b + 1 ramps up while block b drains (on one stream every kernel waits for the complete drain of the one
// before it: ~4 us of a 61 us block).  The caller's stream is NOT made to wait per call (its next `in` would inherit
// that wait and serialise the blocks); it joins the lanes in *_join (DevWindow::join), which every other use of the
// object's window goes through first and which also copies the last block's tail into the object's window.
struct StreamPipe {
    bool on = false;
    // two lanes (a device exposes four hardware queues, the caller's stream and the null stream take two): in the bench
    // three lanes read the same, four lose 8 %
    static constexpr int kLanes = 2;
    hipStream_t lane[kLanes] = {};
    hipEvent_t in = nullptr, done[kLanes] = {};
    bool busy[kLanes] = {};
    unsigned calls = 0;
    const void *prev_tail = nullptr;      // last L samples of the previous pipelined block (null: use the object's window)
    int init() {
        if (lane[0]) return YAGI_OK;
        for (auto &s : lane) YG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        YG_HIP(hipEventCreateWithFlags(&in, hipEventDisableTiming));
        for (auto &e : done) YG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return YAGI_OK;
    }
    ~StreamPipe() {
        for (auto &s : lane)
            if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        if (in) (void)hipEventDestroy(in);
        for (auto &e : done) if (e) (void)hipEventDestroy(e);
    }
};

// ---------------------------------------------------------------------------------------------
// Device window of the last `len` samples + host push queue
// ---------------------------------------------------------------------------------------------
template <class T>
struct DevWindow {
    int len = 0;
    static constexpr int kBufs = 3;   // a ring of three: kernels write window b+1 into next() while block b reads dev()
    DevBuf buf[kBufs];
    int cur = 0;
    // Host mirror for the per-sample calls (push / execute / execute_one: firfilt.rs:220-261, firpfb.rs:255-286,
    // firdecim.rs:179-191).  hbuf[hend - len, hend) = the last len samples, oldest first -- Window<T>::read()'s layout
    // (window.rs:77-85): a push appends, and when the slack behind the window is used up the window moves back to the
    // front.  The two copies are synchronised lazily: a per-sample call after a block call downloads the device window
    // once (ensure_host), a block call after per-sample calls uploads the host window once (ensure_dev); in between,
    // per-sample calls cost a sequential dot product on the host and no launch.
    static constexpr size_t kSlack = 4096;
    std::vector<T> hbuf;
    size_t hend = 0;
    bool host_valid = true, dev_valid = true;
    uint64_t npush = 0;               // samples pushed since the window was made: the ring position of FirFilter's VecDeque
    StreamPipe pipe;                  // pipelined block calls (objects that offer set_pipeline)

    // the caller's stream waits for every block handed to the lanes, then takes over the filter window (the last block's
    // tail); every other access to the window comes through here first
    int join(hipStream_t st) {
        for (int i = 0; i < StreamPipe::kLanes; ++i) {
            if (pipe.busy[i]) YG_HIP(hipStreamWaitEvent(st, pipe.done[i], 0));
            pipe.busy[i] = false;
        }
        if (pipe.prev_tail) {
            YG_TRY(launch_update_window<T>(dev(), static_cast<const T *>(pipe.prev_tail), (size_t)len, len, next(), st));
            flip();
            pipe.prev_tail = nullptr;
        }
        return YAGI_OK;
    }
    // one pipelined block: the caller's stream records `in`, lane (calls & 1) waits for it; returns the lane and the
    // window the block kernel reads.  finish_piped() after the launch.
    int begin_piped(hipStream_t st, hipStream_t *lane, const T **win) {
        const int i = (int)(pipe.calls % (unsigned)StreamPipe::kLanes);
        YG_HIP(hipEventRecord(pipe.in, st));
        YG_HIP(hipStreamWaitEvent(pipe.lane[i], pipe.in, 0));
        *lane = pipe.lane[i];
        *win = pipe.prev_tail ? static_cast<const T *>(pipe.prev_tail) : dev();
        return YAGI_OK;
    }
    int finish_piped(const T *x, size_t n) {          // n >= len
        const int i = (int)(pipe.calls % (unsigned)StreamPipe::kLanes);
        YG_HIP(hipEventRecord(pipe.done[i], pipe.lane[i]));
        pipe.busy[i] = true;
        ++pipe.calls;
        pipe.prev_tail = x + (n - (size_t)len);
        host_valid = false;
        return YAGI_OK;
    }
    bool can_pipe(size_t n) const { return pipe.on && dev_valid && n >= (size_t)len; }

    int init(int n, hipStream_t st) {
        len = n;
        for (auto &b : buf) YG_TRY(b.alloc((size_t)n * sizeof(T)));
        hbuf.assign((size_t)n + kSlack, T{});
        npush = 0;
        return reset(st);
    }
    int reset(hipStream_t st) {       // zeroes the samples in place; the ring position is kept (firfilt.rs:209-213)
        YG_TRY(join(st));
        YG_HIP(hipMemsetAsync(buf[cur].p, 0, (size_t)len * sizeof(T), st));
        std::fill(hbuf.begin(), hbuf.begin() + len, T{});
        hend = (size_t)len;
        host_valid = dev_valid = true;
        return YAGI_OK;
    }
    const T *dev() const { return buf[cur].template as<T>(); }
    // for kernels that write the next window themselves: the other buffer, then flip()
    T *next() { return buf[(cur + 1) % kBufs].template as<T>(); }
    void flip() { cur = (cur + 1) % kBufs; host_valid = false; }
    // window <- last len of (window ++ x_dev[0..n))
    int advance(const T *x_dev, size_t n, hipStream_t st) {
        if (n == 0) return YAGI_OK;
        YG_TRY(launch_update_window<T>(dev(), x_dev, n, len, next(), st));
        flip();
        return YAGI_OK;
    }
    const T *host() const { return hbuf.data() + (hend - (size_t)len); }        // oldest first; valid after ensure_host
    void push(T v) {                                                              // after ensure_host
        if (hend == hbuf.size()) {
            std::memmove(hbuf.data(), hbuf.data() + (hend - (size_t)len + 1), ((size_t)len - 1) * sizeof(T));
            hend = (size_t)len - 1;
        }
        hbuf[hend++] = v;
        ++npush;
        dev_valid = false;
    }
    // the device window is what the block kernels read: bring it up to date with the host mirror
    int ensure_dev(hipStream_t st) {
        YG_TRY(join(st));
        if (dev_valid) return YAGI_OK;
        YG_HIP(hipMemcpyAsync(buf[cur].p, host(), (size_t)len * sizeof(T), hipMemcpyHostToDevice, st));
        YG_HIP(hipStreamSynchronize(st));
        dev_valid = true;
        return YAGI_OK;
    }
    // the host mirror is what the per-sample calls read: fetch the window a block kernel left on the device
    int ensure_host(hipStream_t st) {
        YG_TRY(join(st));
        if (host_valid) return YAGI_OK;
        YG_HIP(hipMemcpyAsync(hbuf.data(), buf[cur].p, (size_t)len * sizeof(T), hipMemcpyDeviceToHost, st));
        YG_HIP(hipStreamSynchronize(st));
        hend = (size_t)len;
        host_valid = true;
        return YAGI_OK;
    }
    int clone_from(const DevWindow &o, hipStream_t st) {       // o.ensure_dev() has run
        YG_TRY(init(o.len, st));
        YG_HIP(hipMemcpyAsync(buf[cur].p, o.buf[o.cur].p, (size_t)len * sizeof(T), hipMemcpyDeviceToDevice, st));
        YG_HIP(hipStreamSynchronize(st));
        npush = o.npush;
        host_valid = false;
        return YAGI_OK;
    }
};

// the per-sample arithmetic (host.cpp): the reference's own sequential sums, products unfused
template <class T, class C> T host_fir_ring_dot(const T *w, size_t L, size_t head, const C *h, C scale);
template <class T, class C> T host_fir_window_dot(const T *w, size_t L, const C *h, C scale);

// workspaces shared by the host-pointer entry points of one object
struct Workspace {
    DevBuf x, y, scratch;
};

// ---------------------------------------------------------------------------------------------
// FirFilter<T,C>
// ---------------------------------------------------------------------------------------------
template <class K>
struct FirFilt {
    using T = typename K::T;
    using C = typename K::C;
    hipStream_t st = nullptr;
    int L = 0, Lp = 0;
    std::vector<C> h;          // host copy (get_coefficients)
    C scale = one_of<C>();
    DevBuf taps;               // h[0..L) on device
    DevBuf taps_pad;           // crcf only: zero-padded to Lp floats for the sliding kernel
    DevBuf apack;              // crcf only, L <= 256: Toeplitz A-operand table of the MFMA kernel
    int Lm = 0;                // padded length of the MFMA form (0 = not available)
    DevBuf hfreq, twf, twb;    // crcf only, L <= 2049: FFT_4096{[h;0]} and both twiddle tables (fast convolution)
    DevBuf gfft, gfft_s;       // crcf only, L <= 257: conj(DFT_512{reversed taps}) / 512 and its scaled copy (fast correction sum)
    DevBuf hfreq_s;            // scale * hfreq for the frequency-domain stream kernel, rebuilt when the scale changes
    bool hfreq_s_valid = false;
    float hfreq_s_scale = 0.f;
    bool conv_ready = false;
    DevWindow<T> w;
    Workspace ws;
    int kernel_choice = 0;     // 0 auto, 1 general, 2 sliding (crcf), 3 MFMA Toeplitz (crcf, L <= 256), 4 fast convolution
    int prepare_conv();

    int load_taps(const C *hh, size_t n) {
        if (n == 0) return fail(YAGI_ERR_CONFIG, "filter length must be greater than zero");
        if (n > (size_t)1 << 24) return fail(YAGI_ERR_CONFIG, "filter too long");
        h.assign(hh, hh + n);
        L = (int)n;
        YG_TRY(taps.alloc(n * sizeof(C)));
        YG_TRY(upload(taps.p, h.data(), n * sizeof(C), st));
        conv_ready = false;
        hfreq_s_valid = false;
        if (K::id == 1) {
            Lp = (L + 31) / 32 * 32;
            std::vector<float> hp((size_t)Lp, 0.0f);
            std::memcpy(hp.data(), h.data(), n * sizeof(float));
            YG_TRY(taps_pad.alloc((size_t)Lp * sizeof(float)));
            YG_TRY(upload(taps_pad.p, hp.data(), (size_t)Lp * sizeof(float), st));
            Lm = mfma_lp_for(L);
            if (Lm) {
                std::vector<float> ap(toeplitz_pack_floats(Lm));
                pack_toeplitz_taps(reinterpret_cast<const float *>(h.data()), L, Lm, ap.data());
                YG_TRY(apack.alloc(ap.size() * sizeof(float)));
                YG_TRY(upload(apack.p, ap.data(), ap.size() * sizeof(float), st));
            }
        }
        return YAGI_OK;
    }
    int init(const C *hh, size_t n) {
        YG_TRY(require_device());
        YG_TRY(load_taps(hh, n));
        return w.init(L, st);
    }
    int block_dev(const T *x, size_t n, T *y);
    // execute() on the host mirror (firfilt.rs:241-246): the VecDeque's head moves back one slot per push, so after
    // npush pushes it sits at (-npush) mod L and as_slices() splits the newest-first sequence L - head from its start
    T host_execute() const {
        const size_t Ls = (size_t)L, head = (Ls - (size_t)(w.npush % Ls)) % Ls;
        return host_fir_ring_dot<T, C>(w.host(), Ls, head, h.data(), scale);
    }
};

// execute_block on device data.  The auto choice is always a direct form (the dotprod sums of the reference,
// exact on integer-valued data); the overlap-save kernel (kernel_choice 4) is opt-in for every type
// combination: 2.5x (crcf 256 taps) to 10x (rrrf / cccf) faster on long blocks, equal to f32 rounding.
// With the pipeline on (set_pipeline; StreamPipe) a block call of at least L samples runs on one of the object's two
// lanes and reads its window from the tail of the previous call's input.
template <class K>
int FirFilt<K>::block_dev(const T *x, size_t n, T *y) {
    if (n == 0) return w.ensure_dev(st);
    const bool piped = w.can_pipe(n) && (kernel_choice != 4 || conv_ready);
    hipStream_t s = st;
    const T *win = nullptr;
    if (piped) YG_TRY(w.begin_piped(st, &s, &win));
    else { YG_TRY(w.ensure_dev(st)); win = w.dev(); }
    T *wnext = piped ? nullptr : w.next();
    w.npush += n;
    const bool conv = kernel_choice == 4 && L <= 2049;
    if (conv) {
        YG_TRY(prepare_conv());
        if constexpr (K::id == 0)
            YG_TRY(launch_fir_rrrf_fftconv(win, x, 0, n, hfreq.as<cf32>(), scale, L, twf.as<cf32>(), twb.as<cf32>(), y, n, s, wnext));
        else
            YG_TRY(launch_fir_cccf_fftconv(win, x, 0, n, hfreq.as<cf32>(), scale, L, twf.as<cf32>(), twb.as<cf32>(), y, n, s, wnext));
    } else {
        YG_TRY((launch_fir_block<K>(win, x, taps.template as<C>(), L, 1, scale, y, n, s, 0, wnext)));
    }
    if (piped) return w.finish_piped(x, n);
    w.flip();                                       // the kernel's last workgroup wrote the next window
    return YAGI_OK;
}
template <>
int FirFilt<CRCF>::block_dev(const cf32 *x, size_t n, cf32 *y) {
    if (n == 0) return w.ensure_dev(st);
    const bool conv = kernel_choice == 4 && L <= 2049;
    // auto: the matrix-pipe Toeplitz form where it exists (<= 256 taps) and the block fills the chip -- the same sums as
    // the sliding vector form (both exact on integer data), but the vector form runs into the chip's power cap on
    // random data (0.63 of the FP32 peak, 0.80 on all-zero input) and the matrix form does not (0.79 either way:
    // profiles/r03_notes.md); else the register-sliding vector form
    const bool mfma = Lm && (kernel_choice == 3 || (kernel_choice == 0 && n >= ((size_t)1 << 16)));
    const bool slide = (kernel_choice == 2) || (kernel_choice == 0 && Lp <= kSlideMaxTaps && n >= 1024);
    const bool piped = w.can_pipe(n) && (!conv || conv_ready);
    hipStream_t s = st;
    const cf32 *win = nullptr;
    if (piped) YG_TRY(w.begin_piped(st, &s, &win));
    else { YG_TRY(w.ensure_dev(st)); win = w.dev(); }
    cf32 *wnext = piped ? nullptr : w.next();
    w.npush += n;
    if (conv) {
        YG_TRY(prepare_conv());
        YG_TRY(launch_fir_crcf_fftconv(win, x, 0, n, hfreq.as<cf32>(), scale, L, twf.as<cf32>(), twb.as<cf32>(), y, n, s, wnext));
    } else if (mfma) {
        YG_TRY(launch_fir_crcf_mfma(win, x, apack.as<float>(), L, Lm, scale, y, n, s, wnext));
    } else if (slide && Lp <= kSlideMaxTaps) {
        YG_TRY(launch_fir_crcf_slide(win, x, taps_pad.as<float>(), L, Lp, scale, y, n, s, wnext));
    } else {
        YG_TRY((launch_fir_block<CRCF>(win, x, taps.as<float>(), L, 1, scale, y, n, s, 0, wnext)));
    }
    if (piped) return w.finish_piped(x, n);
    w.flip();                                       // the kernel's last workgroup wrote the next window
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// FirDecimationFilter<T,C>
// ---------------------------------------------------------------------------------------------
template <class K>
struct FirDecim {
    using T = typename K::T;
    using C = typename K::C;
    hipStream_t st = nullptr;
    int L = 0, M = 1;
    std::vector<C> h;
    C scale = one_of<C>();
    DevBuf taps;
    DevWindow<T> w;
    Workspace ws;

    int init(size_t m, const C *hh, size_t n) {
        if (n == 0) return fail(YAGI_ERR_CONFIG, "filter length must be greater than zero");
        if (m == 0) return fail(YAGI_ERR_CONFIG, "decimation factor must be greater than zero");
        if (n > (size_t)1 << 24 || m > (size_t)1 << 20) return fail(YAGI_ERR_CONFIG, "filter too large");
        YG_TRY(require_device());
        h.assign(hh, hh + n);
        L = (int)n;
        M = (int)m;
        YG_TRY(taps.alloc(n * sizeof(C)));
        YG_TRY(upload(taps.p, h.data(), n * sizeof(C), st));
        return w.init(L, st);
    }
    // n outputs from n*M device samples
    int block_dev(const T *x, size_t n, T *y) {
        YG_TRY(w.ensure_dev(st));
        if (n == 0) return YAGI_OK;
        YG_TRY((launch_fir_block<K>(w.dev(), x, taps.template as<C>(), L, M, scale, y, n, st, 0, w.next())));
        w.flip();                                   // the kernel's last workgroup wrote the next window
        return YAGI_OK;
    }
};

// ---------------------------------------------------------------------------------------------
// FirPfbFilter<T,C>
// ---------------------------------------------------------------------------------------------
template <class K>
struct FirPfb {
    using T = typename K::T;
    using C = typename K::C;
    hipStream_t st = nullptr;
    int nf = 0, Ls = 0;
    std::vector<C> hb;         // [nf][Ls], natural order: hb[i][k] = h[i + k*nf]
    C scale = one_of<C>();
    DevBuf taps;
    DevWindow<T> w;
    Workspace ws;

    int init(size_t num_filters, const C *hh, size_t h_len) {
        if (num_filters == 0) return fail(YAGI_ERR_CONFIG, "number of filters must be greater than zero");
        if (h_len == 0) return fail(YAGI_ERR_CONFIG, "filter length must be greater than zero");
        const size_t hs = h_len / num_filters;            // firpfb.rs:42 (floor)
        if (hs == 0) return fail(YAGI_ERR_CONFIG, "window size must be greater than zero");  // window.rs:14
        if (num_filters > (size_t)1 << 20 || hs > (size_t)1 << 20) return fail(YAGI_ERR_CONFIG, "filter bank too large");
        YG_TRY(require_device());
        nf = (int)num_filters;
        Ls = (int)hs;
        hb.resize((size_t)nf * Ls);
        for (int i = 0; i < nf; ++i)
            for (int k = 0; k < Ls; ++k) hb[(size_t)i * Ls + k] = hh[i + (size_t)k * nf];
        YG_TRY(taps.alloc(hb.size() * sizeof(C)));
        YG_TRY(upload(taps.p, hb.data(), hb.size() * sizeof(C), st));
        return w.init(Ls, st);
    }
    int check_branch(size_t i) const {
        if (i >= (size_t)nf) return fail(YAGI_ERR_CONFIG, "filterbank index (%zu) exceeds maximum (%d)", i, nf);
        return YAGI_OK;
    }
    int block_dev(size_t i, const T *x, size_t n, T *y) {
        YG_TRY(check_branch(i));
        YG_TRY(w.ensure_dev(st));
        YG_TRY((launch_fir_block<K>(w.dev(), x, taps.template as<C>() + i * (size_t)Ls, Ls, 1, scale, y, n, st)));
        return w.advance(x, n, st);
    }
};

// ---------------------------------------------------------------------------------------------
// Fft
// ---------------------------------------------------------------------------------------------
struct FftPlan {
    FftPlanDev d;
    DevBuf tw;
    Workspace ws;
    DevBuf bs_w, bs_bf, bs_twf, bs_twb, bs_scratch;      // Bluestein resources (sizes with a large prime factor)
    std::unique_ptr<FftPlan> bs_fwd, bs_bwd;             // Bluestein over m > 8192: the m-point plans
    std::unique_ptr<FftPlan> fs_p1, fs_p2;               // four-step: the n1- and n2-point plans
    DevBuf fs_scratch, fs_wn, fs_wsplit;
};

// radix list of the mixed-radix kernel: the power of two in as few passes as radix <= 16 allows (bits spread
// evenly), then the odd primes in ascending order (3, 5, 7 have register butterflies, larger ones direct sums)
static void factorize(int n, int *fac, int &nfac) {
    nfac = 0;
    int lg = 0;
    while (n % 2 == 0) { ++lg; n /= 2; }
    if (lg) {
        const int np = (lg + 3) / 4;
        for (int i = 0; i < np; ++i) fac[nfac++] = 1 << (lg / np + (i < lg % np ? 1 : 0));
    }
    for (int p = 3; (long long)p * p <= n; p += 2)
        while (n % p == 0) { fac[nfac++] = p; n /= p; }
    if (n > 1) fac[nfac++] = n;
}

// W_n^m, m in [0,n); with half_too the W_{n/2} table follows at offset n (the 8192-point kernel runs two
// 4096-point transforms and needs both)
static int make_twiddles(int n, int dir, DevBuf &buf, bool half_too = false) {
    std::vector<cf32> t((size_t)n + (half_too ? (size_t)n / 2 : 0));
    const double s = (dir == YAGI_FFT_FORWARD) ? -1.0 : 1.0;
    for (int m = 0; m < n; ++m) {
        const double a = s * 2.0 * M_PI * (double)m / (double)n;
        t[m] = cf32{(float)std::cos(a), (float)std::sin(a)};
    }
    if (half_too)
        for (int m = 0; m < n / 2; ++m) {
            const double a = s * 2.0 * M_PI * (double)m / (double)(n / 2);
            t[(size_t)n + m] = cf32{(float)std::cos(a), (float)std::sin(a)};
        }
    YG_TRY(buf.alloc(t.size() * sizeof(cf32)));
    return upload(buf.p, t.data(), t.size() * sizeof(cf32), nullptr);
}

// twiddle table of the 4096-point stream kernels: W_4096^m (m < 4096, forward) followed by six 256-entry rows the
// frequency-domain kernels read with the lane index (coalesced, no gathers): rows 0-3 W^{t 2^k}, row 4 W^{(t&15)(t>>4)},
// row 5 W_256^{(t&15)(t>>4)}
static int make_stream_twiddles(DevBuf &buf) {
    std::vector<cf32> tab(4096 + 6 * 256);
    auto W = [](unsigned m) {
        const double a = -2.0 * M_PI * (double)(m & 4095u) / 4096.0;
        return cf32{(float)std::cos(a), (float)std::sin(a)};
    };
    for (unsigned m = 0; m < 4096; ++m) tab[m] = W(m);
    for (unsigned t = 0; t < 256; ++t) {
        for (unsigned k = 0; k < 4; ++k) tab[4096 + 256 * k + t] = W(t << k);
        tab[4096 + 1024 + t] = W((t & 15u) * (t >> 4));
        tab[4096 + 1280 + t] = W(16u * (((t & 15u) * (t >> 4)) & 255u));
    }
    YG_TRY(buf.alloc(tab.size() * sizeof(cf32)));
    return upload(buf.p, tab.data(), tab.size() * sizeof(cf32), nullptr);
}

// W_n^m for any m < n as a product of two exact entries: tab[m & 4095] = W_n^{m & 4095}, tab[4096 + (m >> 12)] = W_n^{4096 (m >> 12)}
static int make_split_twiddles(size_t n, int dir, DevBuf &buf) {
    const size_t nhi = (n + 4095) / 4096;
    std::vector<cf32> tab(4096 + nhi);
    const double s = dir == YAGI_FFT_FORWARD ? -1.0 : 1.0;
    for (size_t j = 0; j < 4096; ++j) {
        const double a = s * 2.0 * M_PI * (double)j / (double)n;
        tab[j] = cf32{(float)std::cos(a), (float)std::sin(a)};
    }
    for (size_t j = 0; j < nhi; ++j) {
        const double a = s * 2.0 * M_PI * (double)(4096 * j) / (double)n;
        tab[4096 + j] = cf32{(float)std::cos(a), (float)std::sin(a)};
    }
    YG_TRY(buf.alloc(tab.size() * sizeof(cf32)));
    return upload(buf.p, tab.data(), tab.size() * sizeof(cf32), nullptr);
}

static int fft_plan_init(FftPlan &p, size_t n, int dir) {
    if (n == 0) return fail(YAGI_ERR_CONFIG, "fft length must be greater than zero");
    if (dir != YAGI_FFT_FORWARD && dir != YAGI_FFT_BACKWARD) return fail(YAGI_ERR_CONFIG, "bad fft direction");
    const bool pow2 = (n & (n - 1)) == 0;
    if (n > kFftMaxPow2 || (!pow2 && 2 * n - 1 > kFftMaxPow2))
        return fail(YAGI_ERR_CONFIG, "fft length %zu not supported (powers of two up to %zu, other sizes up to %zu)", n,
                    kFftMaxPow2, kFftMaxPow2 / 2);
    YG_TRY(require_device());
    p.d.n = (int)n;
    p.d.dir = dir;
    if (n > (size_t)kFftMaxLds) {
        // four-step: n = n1 n2, both <= 8192, as balanced as the divisors of n allow, and neither with a prime factor
        // that would send it to Bluestein (> 89); a size without such a split (a large prime factor) takes Bluestein
        size_t n2 = 0;
        if (pow2 && n >= 65536) {
            // n = 256 n2: 256-point columns in registers, then the n2-point rows (fft_kernels.hip: fft_tile256_kernel)
            n2 = n / 256;
            p.fs_p1 = std::make_unique<FftPlan>();
            p.fs_p2 = std::make_unique<FftPlan>();
            YG_TRY(fft_plan_init(*p.fs_p1, 256, dir));
            YG_TRY(fft_plan_init(*p.fs_p2, n2, dir));
            size_t chunk = ((size_t)1 << 23) / n;              // 64 MiB per scratch buffer (smaller chunks measured slower)
            if (chunk < 1) chunk = 1;
            YG_TRY(p.fs_scratch.alloc((n2 == 256 ? 1 : 2) * chunk * n * sizeof(cf32)));
            YG_TRY(make_split_twiddles(n, dir, p.fs_wsplit));
            p.d.fs_n1 = 256;
            p.d.fs_n2 = (int)n2;
            p.d.fs_p1 = &p.fs_p1->d;
            p.d.fs_p2 = &p.fs_p2->d;
            p.d.fs_scratch = p.fs_scratch.as<cf32>();
            p.d.fs_chunk = (int)chunk;
            p.d.fs_wlo = p.fs_wsplit.as<cf32>();
            p.d.fs_whi = p.fs_wsplit.as<cf32>() + 4096;
            return YAGI_OK;
        }
        auto smooth = [](size_t v) {
            for (size_t q = 2; q <= 89 && v > 1; ++q) while (v % q == 0) v /= q;
            return v == 1;
        };
        for (size_t d = (size_t)std::sqrt((double)n) + 1; d >= 2; --d)
            if (n % d == 0 && n / d <= (size_t)kFftMaxLds && d <= (size_t)kFftMaxLds && smooth(d) && smooth(n / d)) { n2 = d; break; }
        if (n2) {
            const size_t n1 = n / n2;
            p.fs_p1 = std::make_unique<FftPlan>();
            p.fs_p2 = std::make_unique<FftPlan>();
            YG_TRY(fft_plan_init(*p.fs_p1, n1, dir));
            YG_TRY(fft_plan_init(*p.fs_p2, n2, dir));
            size_t chunk = ((size_t)1 << 23) / n;              // 2 x 64 MiB of scratch
            if (chunk < 1) chunk = 1;
            YG_TRY(p.fs_scratch.alloc(2 * chunk * n * sizeof(cf32)));
            p.d.fs_n1 = (int)n1;
            p.d.fs_n2 = (int)n2;
            p.d.fs_p1 = &p.fs_p1->d;
            p.d.fs_p2 = &p.fs_p2->d;
            p.d.fs_scratch = p.fs_scratch.as<cf32>();
            p.d.fs_chunk = (int)chunk;
            YG_TRY(make_split_twiddles(n, dir, p.fs_wsplit));
            p.d.fs_wlo4 = p.fs_wsplit.as<cf32>();
            p.d.fs_whi4 = p.fs_wsplit.as<cf32>() + 4096;
            auto small_pow2 = [](size_t v) { return v >= 64 && v <= 256 && (v & (v - 1)) == 0; };
            if (small_pow2(n1) && small_pow2(n2)) {            // two-launch form (fft_kernels.hip: fft_twopass_kernel)
                YG_TRY(make_twiddles((int)n, dir, p.fs_wn));
                p.d.fs_wn = p.fs_wn.as<cf32>();
            }
            return YAGI_OK;
        }
        if (pow2) return fail(YAGI_ERR_INTERNAL, "no four-step split for %zu", n);
    }
    int maxp = 1;
    if (n <= (size_t)kFftMaxLds) {
        factorize((int)n, p.d.fac, p.d.nfac);
        YG_TRY(make_twiddles((int)n, dir, p.tw, n == 8192));
        p.d.tw = p.tw.as<cf32>();
        for (int i = 0; i < p.d.nfac; ++i) maxp = p.d.fac[i] > maxp ? p.d.fac[i] : maxp;
    }
    // A prime factor p costs O(n p) per transform in the mixed-radix kernel's direct-sum pass; beyond 89 (measured
    // crossover) -- and for every size that does not fit the one-kernel paths -- Bluestein's chirp-z form is used:
    //   X[k] = w[k] sum_j (x[j] w[j]) conj(w[k-j]),  w[k] = e^{-+ j pi k^2 / n}
    static const int bs_minp = getenv("YAGI_HIP_BLUESTEIN_MIN_PRIME") ? atoi(getenv("YAGI_HIP_BLUESTEIN_MIN_PRIME")) : 89;
    if (n > (size_t)kFftMaxLds || maxp > bs_minp) {
        size_t m = 1;
        while (m < 2 * n - 1) m *= 2;
        if (m < 256) m = 256;
        const double sgn = (dir == YAGI_FFT_FORWARD) ? -1.0 : 1.0;
        std::vector<cf32> w(n), b(m, cf32{0.f, 0.f});
        for (size_t k = 0; k < n; ++k) {
            const unsigned long long k2 = ((unsigned long long)k * k) % (2ull * n);      // exact phase index
            const double a = sgn * M_PI * (double)k2 / (double)n;
            w[k] = cf32{(float)std::cos(a), (float)std::sin(a)};
            const cf32 cw{w[k].re, -w[k].im};
            b[k] = cw;
            if (k) b[m - k] = cw;
        }
        YG_TRY(p.bs_w.alloc(n * sizeof(cf32)));
        YG_TRY(upload(p.bs_w.p, w.data(), n * sizeof(cf32), nullptr));
        // 2 x 64 MiB of scratch (more for one big transform).  With 2 x 16 MiB the five-stage form was launch-bound on long
        // batches (n = 12 289: 2400 launches of ~9 us for 2^28 points): 0.20 -> 0.30 TB/s, n = 509: 0.42 -> 0.59; 2 x 128 MiB the same
        static const int bs_lg = getenv("YAGI_HIP_BS_CHUNK_LOG2") ? atoi(getenv("YAGI_HIP_BS_CHUNK_LOG2")) : 23;
        size_t chunk = ((size_t)1 << (bs_lg < 16 ? 16 : (bs_lg > 26 ? 26 : bs_lg))) / m;
        if (chunk < 1) chunk = 1;
        if (m == 4096 || m == 8192) chunk = 1;                 // bluestein_fused_kernel: the scratch only serves FFT_m{b} below
        YG_TRY(p.bs_scratch.alloc(2 * chunk * m * sizeof(cf32)));
        YG_TRY(p.bs_bf.alloc(m * sizeof(cf32)));
        FftPlanDev f;                                          // FFT_m of the chirp filter by the device transform itself
        if (m > (size_t)kFftMaxLds) {
            p.bs_fwd = std::make_unique<FftPlan>();
            p.bs_bwd = std::make_unique<FftPlan>();
            YG_TRY(fft_plan_init(*p.bs_fwd, m, YAGI_FFT_FORWARD));
            YG_TRY(fft_plan_init(*p.bs_bwd, m, YAGI_FFT_BACKWARD));
            p.d.bs_fwd = &p.bs_fwd->d;
            p.d.bs_bwd = &p.bs_bwd->d;
            f = p.bs_fwd->d;
        } else {
            YG_TRY(make_twiddles((int)m, YAGI_FFT_FORWARD, p.bs_twf, m == 8192));
            YG_TRY(make_twiddles((int)m, YAGI_FFT_BACKWARD, p.bs_twb, m == 8192));
            f.n = (int)m;
            f.dir = YAGI_FFT_FORWARD;
            f.tw = p.bs_twf.as<cf32>();
        }
        YG_TRY(upload(p.bs_scratch.p, b.data(), m * sizeof(cf32), nullptr));
        YG_TRY(launch_fft_batch(f, p.bs_scratch.as<cf32>(), p.bs_bf.as<cf32>(), 1, nullptr));
        YG_HIP(hipStreamSynchronize(nullptr));
        p.d.bs_m = (int)m;
        p.d.bs_w = p.bs_w.as<cf32>();
        p.d.bs_bf = p.bs_bf.as<cf32>();
        p.d.bs_twf = p.bs_twf.as<cf32>();
        p.d.bs_twb = p.bs_twb.as<cf32>();
        p.d.bs_scratch = p.bs_scratch.as<cf32>();
        p.d.bs_chunk = (int)chunk;
    }
    return YAGI_OK;
}

// fast-convolution resources of a filter: FFT_4096{[h;0]} and both twiddle tables
template <class K>
int FirFilt<K>::prepare_conv() {
    if (conv_ready) return YAGI_OK;
    if (L > 2049) return fail(YAGI_ERR_CONFIG, "fast convolution needs <= 2049 taps");
    YG_TRY(make_stream_twiddles(twf));      // W_4096 forward table + the lane-indexed rows of the table-twiddle transform
    YG_TRY(make_twiddles(4096, YAGI_FFT_BACKWARD, twb));
    // FFT_4096{[h; 0]} evaluated in double on the host (L x 4096 terms, once per tap set), rounded once
    {
        std::vector<double> cs(4096), sn(4096);
        for (int m = 0; m < 4096; ++m) {
            const double a = -2.0 * M_PI * (double)m / 4096.0;
            cs[m] = std::cos(a);
            sn[m] = std::sin(a);
        }
        constexpr bool ctaps = sizeof(C) == sizeof(cf32);
        const float *hf = reinterpret_cast<const float *>(h.data());      // L floats, or L {re, im} pairs
        std::vector<cf32> hp(4096);
        for (int k = 0; k < 4096; ++k) {
            double re = 0.0, im = 0.0;
            for (int i = 0; i < L; ++i) {
                const int m = (int)(((long long)i * k) & 4095);
                const double hr = ctaps ? hf[2 * i] : hf[i], hi = ctaps ? hf[2 * i + 1] : 0.0;
                re += hr * cs[m] - hi * sn[m];
                im += hr * sn[m] + hi * cs[m];
            }
            hp[k] = cf32{(float)re, (float)im};
        }
        YG_TRY(hfreq.alloc(4096 * sizeof(cf32)));
        YG_TRY(upload(hfreq.p, hp.data(), 4096 * sizeof(cf32), st));
        // reversed taps g[j] = h[L-1-j] of the frame-boundary correction (freq_kernels.hip)
        if (K::id == 1 && L <= 257) {
            std::vector<float> g(256, 0.0f);
            for (int j = 0; j < L - 1; ++j) g[j] = hf[L - 1 - j];
            // the correction sum as a 512-point circular correlation: conj(DFT_512{g}) / 512 (freq_kernels.hip)
            std::vector<cf32> gf(512);
            for (int k = 0; k < 512; ++k) {
                double re = 0.0, im = 0.0;
                for (int j = 0; j < L - 1; ++j) {
                    const double a = -2.0 * M_PI * (double)((j * k) & 511) / 512.0;
                    re += (double)g[j] * std::cos(a);
                    im += (double)g[j] * std::sin(a);
                }
                gf[k] = cf32{(float)(re / 512.0), (float)(-im / 512.0)};
            }
            YG_TRY(gfft.alloc(512 * sizeof(cf32)));
            YG_TRY(upload(gfft.p, gf.data(), 512 * sizeof(cf32), st));
        }
        YG_HIP(hipStreamSynchronize(st));      // the host vectors go out of scope
    }
    conv_ready = true;
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// fused firfilt_crcf -> FFT stream
// ---------------------------------------------------------------------------------------------
struct FirFft {
    FirFilt<CRCF> fir;
    size_t nfft = 0;
    DevBuf tw;
    FftPlan plan;              // nfft != 4096: overlap-save FIR, then this plan over the frames
    int variant = 0;
    DevBuf xin, yout;
    DevBuf scratch;            // variant 3: the FIR output stream between the two kernels
    int join() { return fir.w.join(fir.st); }
};

// ---------------------------------------------------------------------------------------------
// channelizers
// ---------------------------------------------------------------------------------------------
struct PfbCh {
    hipStream_t st = nullptr;
    int M = 0, p = 0;
    DevBuf h, tw;
    DevWindow<cf32> hist;      // (p-1)*M samples (at least 1 kept so the buffers exist)
    DevWindow<cf32> syn_hist;  // synthesizer: the last (p-1)*M channel samples (its own state, like liquid's separate objects)
    Workspace ws;
};
struct PfbCh2 {
    hipStream_t st = nullptr;
    int M = 0, m = 0;
    DevBuf h, tw;
    DevWindow<cf32> hist;      // (2m-1)*M + M/2 samples
    uint64_t step = 0;
    DevWindow<cf32> syn_hist;  // synthesizer: the last (4m-1)*M channel samples; its own step parity
    uint64_t syn_step = 0;
    Workspace ws;
    DevBuf shard, gathered;    // sharded analyzer: this rank's sub-bands, and every rank's ([rank][step][M/R] per chunk)
};

}  // namespace yagi

using namespace yagi;

#define CHECK_Q(q)                                                                              \
    do {                                                                                        \
        if (!(q)) return fail(YAGI_ERR_CONFIG, "null handle");                                  \
    } while (0)
// The block kernels read x (tile halos, the window after the block) while other workgroups already store y: an
// in-place or overlapping call would corrupt outputs AND the carried state.  Rust's borrow rules make it unwritable in
// the reference (&[T] and &mut [T] cannot alias); the C ABI says so explicitly.
static int check_noalias(const void *x, size_t xbytes, const void *y, size_t ybytes) {
    const char *a = static_cast<const char *>(x), *b = static_cast<const char *>(y);
    if (a < b + ybytes && b < a + xbytes)
        return fail(YAGI_ERR_CONFIG, "input and output buffers overlap (in-place execution is not supported)");
    return YAGI_OK;
}
#define CHECK_NOALIAS(x, nx, y, ny) YG_TRY(check_noalias((x), (size_t)(nx) * sizeof(*(x)), (y), (size_t)(ny) * sizeof(*(y))))
#define CHECK_PTR(p)                                                                            \
    do {                                                                                        \
        if (!(p)) return fail(YAGI_ERR_CONFIG, "null pointer argument");                        \
    } while (0)

extern "C" {

// ---- device plumbing ------------------------------------------------------------------------
int yagi_hip_device_count(int *count) try {
    CHECK_PTR(count);
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return fail(YAGI_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_set_device(int device) try { YG_HIP(hipSetDevice(device)); return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_malloc(void **p, size_t bytes) try {
    CHECK_PTR(p);
    YG_TRY(require_device());
    YG_HIP(hipMalloc(p, bytes ? bytes : 16));
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_free(void *p) try { if (p) YG_HIP(hipFree(p)); return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_memcpy_h2d(void *d, const void *s, size_t n) try { YG_HIP(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_memcpy_d2h(void *d, const void *s, size_t n) try { YG_HIP(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_memset_dev(void *d, int v, size_t n) try { YG_HIP(hipMemset(d, v, n)); return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_device_synchronize(void) try { YG_HIP(hipDeviceSynchronize()); return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_stream_synchronize(yagi_stream_t s) try { YG_HIP(hipStreamSynchronize(to_stream(s))); return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }

int yagi_hip_gen_real_dev(uint64_t seed, uint64_t first, size_t n, float *x, yagi_stream_t s) try {
    YG_TRY(require_device());
    return launch_gen_real(seed, first, n, x, to_stream(s));
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_gen_complex_dev(uint64_t seed, uint64_t first, size_t n, yagi_cf32 *x, yagi_stream_t s) try {
    YG_TRY(require_device());
    return launch_gen_complex(seed, first, n, x, to_stream(s));
} catch (...) { return ::yagi::api_exception(); }

}  // extern "C"

// ---- dotprod ------------------------------------------------------------------------------------
template <class A, class B, class O>
static int dotprod_dev(const A *a, const B *b, size_t n, O *y, hipStream_t st) {
    YG_TRY(require_device());
    const size_t np = dotprod_num_partials(n);
    if (np == 1) return launch_dotprod<A, B, O, float>(a, b, n, false, 1.0f, y, y, st);
    // multi-workgroup case needs scratch for the partials: one grow-only buffer per host thread and device stream
    // user, kept for the life of the process (a per-call hipMalloc / hipFree pair costs more than the reduction of
    // 2^24 elements and forces a device synchronisation); calls on one stream are ordered, so reuse is safe there,
    // and concurrent streams of one thread are serialised by the event below
    // One scratch per host thread AND device (hipGetDevice): the buffer and its event belong to the device that was
    // current when they were made; after yagi_hip_set_device(other) the thread gets that device's own pair.  Released
    // at thread exit.
    struct Scratch { void *p = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; };
    struct PerThread {
        std::vector<std::pair<int, Scratch>> dev;
        ~PerThread() {
            for (auto &e : dev) {
                int cur = -1;
                if (hipGetDevice(&cur) != hipSuccess) return;          // runtime already gone at process exit
                if (hipSetDevice(e.first) != hipSuccess) continue;
                if (e.second.done) (void)hipEventDestroy(e.second.done);
                if (e.second.p) (void)hipFree(e.second.p);
                (void)hipSetDevice(cur);
            }
        }
    };
    static thread_local PerThread pt;
    int devid = 0;
    YG_HIP(hipGetDevice(&devid));
    Scratch *scp = nullptr;
    for (auto &e : pt.dev) if (e.first == devid) scp = &e.second;
    if (!scp) { pt.dev.emplace_back(devid, Scratch{}); scp = &pt.dev.back().second; }
    Scratch &sc = *scp;
    const size_t need = np * sizeof(O);
    if (sc.done) YG_HIP(hipStreamWaitEvent(st, sc.done, 0));           // previous user of the scratch (any stream)
    else YG_HIP(hipEventCreateWithFlags(&sc.done, hipEventDisableTiming));
    if (need > sc.bytes) {
        if (sc.p) { YG_HIP(hipDeviceSynchronize()); YG_HIP(hipFree(sc.p)); sc.p = nullptr; sc.bytes = 0; }
        YG_HIP(hipMalloc(&sc.p, need + need / 2));
        sc.bytes = need + need / 2;
    }
    YG_TRY((launch_dotprod<A, B, O, float>(a, b, n, false, 1.0f, static_cast<O *>(sc.p), y, st)));
    YG_HIP(hipEventRecord(sc.done, st));
    return YAGI_OK;
}
template <class A, class B, class O>
static int dotprod_host(const A *a, const B *b, size_t n, O *y) {
    CHECK_PTR(y);
    if (n && (!a || !b)) return fail(YAGI_ERR_CONFIG, "null pointer argument");
    YG_TRY(require_device());
    DevBuf da, db, dy;
    YG_TRY(da.alloc(n * sizeof(A)));
    YG_TRY(db.alloc(n * sizeof(B)));
    YG_TRY(dy.alloc(sizeof(O)));
    YG_TRY(upload(da.p, a, n * sizeof(A), nullptr));
    YG_TRY(upload(db.p, b, n * sizeof(B), nullptr));
    YG_TRY((dotprod_dev<A, B, O>(da.as<A>(), db.as<B>(), n, dy.as<O>(), nullptr)));
    return download(y, dy.p, sizeof(O), nullptr);
}

extern "C" {
int yagi_hip_dotprod_rrrf(const float *a, const float *b, size_t n, float *y) try { return dotprod_host<float, float, float>(a, b, n, y); } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_dotprod_rccf(const float *a, const yagi_cf32 *b, size_t n, yagi_cf32 *y) try { return dotprod_host<float, cf32, cf32>(a, b, n, y); } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_dotprod_crcf(const yagi_cf32 *a, const float *b, size_t n, yagi_cf32 *y) try { return dotprod_host<cf32, float, cf32>(a, b, n, y); } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_dotprod_cccf(const yagi_cf32 *a, const yagi_cf32 *b, size_t n, yagi_cf32 *y) try { return dotprod_host<cf32, cf32, cf32>(a, b, n, y); } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_dotprod_rrrf_dev(const float *a, const float *b, size_t n, float *y, yagi_stream_t s) try { return dotprod_dev<float, float, float>(a, b, n, y, to_stream(s)); } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_dotprod_rccf_dev(const float *a, const yagi_cf32 *b, size_t n, yagi_cf32 *y, yagi_stream_t s) try { return dotprod_dev<float, cf32, cf32>(a, b, n, y, to_stream(s)); } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_dotprod_crcf_dev(const yagi_cf32 *a, const float *b, size_t n, yagi_cf32 *y, yagi_stream_t s) try { return dotprod_dev<cf32, float, cf32>(a, b, n, y, to_stream(s)); } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_dotprod_cccf_dev(const yagi_cf32 *a, const yagi_cf32 *b, size_t n, yagi_cf32 *y, yagi_stream_t s) try { return dotprod_dev<cf32, cf32, cf32>(a, b, n, y, to_stream(s)); } catch (...) { return ::yagi::api_exception(); }
}

// ---- FIR family: generic bodies, instantiated per type combination by YAGI_FIR_IMPL ----------------
namespace yagi {

template <class K>
static int firfilt_block_host(FirFilt<K> *q, const typename K::T *x, size_t nx, typename K::T *y, size_t ny) {
    using T = typename K::T;
    if (nx != ny) return fail(YAGI_ERR_CONFIG, "input and output block lengths must be equal");
    if (nx == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    YG_TRY(q->ws.x.ensure(nx * sizeof(T)));
    YG_TRY(q->ws.y.ensure(nx * sizeof(T)));
    YG_TRY(q->w.join(q->st));                            // the workspace is reused: host-pointer calls never pipeline
    YG_TRY(upload(q->ws.x.p, x, nx * sizeof(T), q->st));
    const bool was = q->w.pipe.on;
    q->w.pipe.on = false;
    const int rc = q->block_dev(q->ws.x.template as<T>(), nx, q->ws.y.template as<T>());
    q->w.pipe.on = was;
    YG_TRY(rc);
    return download(y, q->ws.y.p, nx * sizeof(T), q->st);
}

template <class K>
static int firdecim_block_host(FirDecim<K> *q, const typename K::T *x, size_t nx, size_t n, typename K::T *y) {
    using T = typename K::T;
    if (n == 0) return YAGI_OK;
    if (nx / (size_t)q->M < n) return fail(YAGI_ERR_CONFIG, "input block too short: need %zu samples, got %zu", n * (size_t)q->M, nx);
    CHECK_PTR(x);
    CHECK_PTR(y);
    const size_t nin = n * (size_t)q->M;
    YG_TRY(q->ws.x.ensure(nin * sizeof(T)));
    YG_TRY(q->ws.y.ensure(n * sizeof(T)));
    YG_TRY(upload(q->ws.x.p, x, nin * sizeof(T), q->st));
    YG_TRY(q->block_dev(q->ws.x.template as<T>(), n, q->ws.y.template as<T>()));
    return download(y, q->ws.y.p, n * sizeof(T), q->st);
}

template <class K>
static int firpfb_block_host(FirPfb<K> *q, size_t i, const typename K::T *x, size_t nx, typename K::T *y, size_t ny) {
    using T = typename K::T;
    YG_TRY(q->check_branch(i));
    const size_t n = nx < ny ? nx : ny;       // zip() stops at the shorter slice (firpfb.rs:296)
    if (n == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    YG_TRY(q->ws.x.ensure(n * sizeof(T)));
    YG_TRY(q->ws.y.ensure(n * sizeof(T)));
    YG_TRY(upload(q->ws.x.p, x, n * sizeof(T), q->st));
    YG_TRY(q->block_dev(i, q->ws.x.template as<T>(), n, q->ws.y.template as<T>()));
    return download(y, q->ws.y.p, n * sizeof(T), q->st);
}

}  // namespace yagi

// ComplexNotch (firfilt.rs:17-44): real coefficients design the notch at f0 directly (a notch pair at +-f0);
// complex coefficients design the DC blocker and mix it up to f0 (a single notch)
static int notch_taps(size_t m, float as_, float f0, std::vector<float> &h) { return design_notch(m, f0, as_, h); }
static int notch_taps(size_t m, float as_, float f0, std::vector<cf32> &h) {
    std::vector<float> hf;
    YG_TRY(design_notch(m, 0.0f, as_, hf));
    h.resize(hf.size());
    for (size_t i = 0; i < hf.size(); ++i) {
        const float phi = 2.0f * 3.14159265358979323846f * f0 * ((float)i - (float)m);
        h[i] = cf32{hf[i] * std::cos(phi), hf[i] * std::sin(phi)};
    }
    return YAGI_OK;
}

// freqresponse (design/mod.rs:666-675): H(fc) = sum_i h[i] e^{-j 2 pi fc i}, the phasor in f64 rounded to f32, the sum in
// Complex32; fir_group_delay (:687-704): Re(sum_i i h[i] e^{+j 2 pi fc i} / sum_i h[i] e^{+j 2 pi fc i}) on the real parts
static cf32 cx_val(float v) { return cf32{v, 0.0f}; }
static cf32 cx_val(cf32 v) { return v; }
template <class C>
static cf32 taps_freqresponse(const std::vector<C> &h, float fc) {
    cf32 acc{0.0f, 0.0f};
    for (size_t i = 0; i < h.size(); ++i) {
        const double a = -2.0 * M_PI * (double)fc * (double)i;
        const cf32 e{(float)std::cos(a), (float)std::sin(a)}, v = cx_val(h[i]);
        acc.re += v.re * e.re - v.im * e.im;
        acc.im += v.re * e.im + v.im * e.re;
    }
    return acc;
}
static cf32 cx_mul(cf32 a, cf32 b) { return cf32{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <class C>
static int taps_groupdelay(const std::vector<C> &h, float fc, float *out) {
    if (h.empty()) return fail(YAGI_ERR_CONFIG, "fir_group_delay(), length must be greater than zero");
    if (fc < -0.5f || fc > 0.5f) return fail(YAGI_ERR_CONFIG, "fir_group_delay(), _fc must be in [-0.5,0.5]");
    cf32 t0{0.f, 0.f}, t1{0.f, 0.f};
    for (size_t i = 0; i < h.size(); ++i) {
        const float a = 2.0f * 3.14159265358979323846f * fc * (float)i, hr = cx_val(h[i]).re;
        const cf32 e{std::cos(a), std::sin(a)};
        t0.re += hr * e.re * (float)i;
        t0.im += hr * e.im * (float)i;
        t1.re += hr * e.re;
        t1.im += hr * e.im;
    }
    const float den = t1.re * t1.re + t1.im * t1.im;
    *out = (t0.re * t1.re + t0.im * t1.im) / den;
    return YAGI_OK;
}

#define YAGI_FIR_IMPL(K, KT, T, C)                                                                  \
    struct yagi_hip_firfilt_##K##_s : FirFilt<KT> {};                                               \
    struct yagi_hip_firdecim_##K##_s : FirDecim<KT> {};                                             \
    struct yagi_hip_firpfb_##K##_s : FirPfb<KT> {};                                                 \
    extern "C" {                                                                                    \
    int yagi_hip_firfilt_##K##_create(const C *h, size_t h_len, yagi_hip_firfilt_##K *q) try {      \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (h_len && !h) return fail(YAGI_ERR_CONFIG, "null pointer argument");                     \
        auto o = std::make_unique<yagi_hip_firfilt_##K##_s>();                                      \
        YG_TRY(o->init(h, h_len));                                                                  \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_create_kaiser(size_t n, float fc, float as_, float mu,               \
                                             yagi_hip_firfilt_##K *q) try {                         \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        std::vector<float> hf;                                                                      \
        YG_TRY(design_kaiser(n, fc, as_, mu, hf));                                                  \
        std::vector<C> hc(hf.size());                                                               \
        for (size_t i = 0; i < hf.size(); ++i) hc[i] = to_c(hf[i], (C *)nullptr);                   \
        return yagi_hip_firfilt_##K##_create(hc.data(), hc.size(), q);                              \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_create_rect(size_t n, yagi_hip_firfilt_##K *q) try {                 \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (n == 0 || n > 1024) return fail(YAGI_ERR_CONFIG, "filter length must be in [1,1024]");  \
        std::vector<C> hc(n, one_of<C>());                                                          \
        return yagi_hip_firfilt_##K##_create(hc.data(), hc.size(), q);                              \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_create_notch(size_t m, float as_, float f0, yagi_hip_firfilt_##K *q) try { \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        std::vector<C> hc;                                                                          \
        YG_TRY(notch_taps(m, as_, f0, hc));                                                         \
        return yagi_hip_firfilt_##K##_create(hc.data(), hc.size(), q);                              \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_create_dc_blocker(size_t m, float as_, yagi_hip_firfilt_##K *q) try { \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        std::vector<float> hf;                                                                      \
        YG_TRY(design_notch(m, 0.0f, as_, hf));                                                     \
        std::vector<C> hc(hf.size());                                                               \
        for (size_t i = 0; i < hf.size(); ++i) hc[i] = to_c(hf[i], (C *)nullptr);                   \
        return yagi_hip_firfilt_##K##_create(hc.data(), hc.size(), q);                              \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_destroy(yagi_hip_firfilt_##K q) try {                                \
        delete q;                                                                                   \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_clone(yagi_hip_firfilt_##K q, yagi_hip_firfilt_##K *out) try {       \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(out);                                                                             \
        *out = nullptr;                                                                             \
        YG_TRY(q->w.ensure_dev(q->st));                                                                  \
        auto o = std::make_unique<yagi_hip_firfilt_##K##_s>();                                      \
        o->st = q->st;                                                                              \
        YG_TRY(o->init(q->h.data(), q->h.size()));                                                  \
        o->scale = q->scale;                                                                        \
        o->kernel_choice = q->kernel_choice;                                                        \
        YG_TRY(o->w.clone_from(q->w, q->st));                                                       \
        *out = o.release();                                                                         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_set_stream(yagi_hip_firfilt_##K q, yagi_stream_t s) try {            \
        CHECK_Q(q);                                                                                 \
        if (q->st == to_stream(s)) return YAGI_OK;                                                  \
        YG_TRY(q->w.join(q->st));                                                                   \
        YG_HIP(hipStreamSynchronize(q->st));                                                        \
        q->st = to_stream(s);                                                                       \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_set_pipeline(yagi_hip_firfilt_##K q, int on) try {                   \
        CHECK_Q(q);                                                                                 \
        YG_TRY(q->w.join(q->st));                                                                   \
        if (on) YG_TRY(q->w.pipe.init());                                                           \
        q->w.pipe.on = on != 0;                                                                     \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_join(yagi_hip_firfilt_##K q) try {                                   \
        CHECK_Q(q);                                                                                 \
        return q->w.join(q->st);                                                                    \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_set_coefficients(yagi_hip_firfilt_##K q, const C *h, size_t n) try { \
        CHECK_Q(q);                                                                                 \
        if (n && !h) return fail(YAGI_ERR_CONFIG, "null pointer argument");                         \
        YG_TRY(q->w.join(q->st));                                                                   \
        YG_HIP(hipStreamSynchronize(q->st));                                                        \
        const bool resize = (n != (size_t)q->L);                                                    \
        YG_TRY(q->load_taps(h, n));                                                                 \
        if (resize) return q->w.init(q->L, q->st);                                                  \
        return q->w.reset(q->st);                                                                   \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_reset(yagi_hip_firfilt_##K q) try {                                  \
        CHECK_Q(q);                                                                                 \
        return q->w.reset(q->st);                                                                   \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_push(yagi_hip_firfilt_##K q, T x) try {                              \
        CHECK_Q(q);                                                                                 \
        YG_TRY(q->w.ensure_host(q->st));                                                            \
        q->w.push(x);                                                                               \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_write(yagi_hip_firfilt_##K q, const T *x, size_t n) try {            \
        CHECK_Q(q);                                                                                 \
        if (n && !x) return fail(YAGI_ERR_CONFIG, "null pointer argument");                         \
        YG_TRY(q->w.ensure_host(q->st));                                                            \
        for (size_t i = 0; i < n; ++i) q->w.push(x[i]);                                             \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_execute(yagi_hip_firfilt_##K q, T *y) try {                          \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(y);                                                                               \
        YG_TRY(q->w.ensure_host(q->st));                                                            \
        *y = q->host_execute();                                                                     \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_execute_one(yagi_hip_firfilt_##K q, T x, T *y) try {                 \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(y);                                                                               \
        YG_TRY(q->w.ensure_host(q->st));                                                            \
        q->w.push(x);                                                                               \
        *y = q->host_execute();                                                                     \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_execute_block(yagi_hip_firfilt_##K q, const T *x, size_t nx, T *y,   \
                                             size_t ny) try {                                       \
        CHECK_Q(q);                                                                                 \
        return firfilt_block_host<KT>(q, x, nx, y, ny);                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_execute_block_dev(yagi_hip_firfilt_##K q, const T *x, size_t n,      \
                                                 T *y) try {                                        \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, n, y, n);                                                                  \
        return q->block_dev(x, n, y);                                                               \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_set_scale(yagi_hip_firfilt_##K q, C s) try {                         \
        CHECK_Q(q);                                                                                 \
        q->scale = s;                                                                               \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_get_scale(yagi_hip_firfilt_##K q, C *s) try {                        \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(s);                                                                               \
        *s = q->scale;                                                                              \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_get_length(yagi_hip_firfilt_##K q, size_t *n) try {                  \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(n);                                                                               \
        *n = (size_t)q->L;                                                                          \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_get_coefficients(yagi_hip_firfilt_##K q, C *h, size_t n) try {       \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(h);                                                                               \
        if (n < (size_t)q->L) return fail(YAGI_ERR_CONFIG, "coefficient buffer too short");         \
        std::memcpy(h, q->h.data(), (size_t)q->L * sizeof(C));                                      \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_freqresponse(yagi_hip_firfilt_##K q, float fc, yagi_cf32 *H) try {   \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(H);                                                                               \
        *H = cx_mul(taps_freqresponse(q->h, fc), cx_val(q->scale));                                 \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firfilt_##K##_groupdelay(yagi_hip_firfilt_##K q, float fc, float *delay) try {     \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(delay);                                                                           \
        return taps_groupdelay(q->h, fc, delay);                                                    \
    } catch (...) { return ::yagi::api_exception(); }                                               \
                                                                                                    \
    int yagi_hip_firdecim_##K##_create(size_t M, const C *h, size_t h_len,                          \
                                       yagi_hip_firdecim_##K *q) try {                              \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (h_len && !h) return fail(YAGI_ERR_CONFIG, "null pointer argument");                     \
        auto o = std::make_unique<yagi_hip_firdecim_##K##_s>();                                     \
        YG_TRY(o->init(M, h, h_len));                                                               \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_create_kaiser(size_t M, size_t m, float as_,                        \
                                              yagi_hip_firdecim_##K *q) try {                       \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (M < 2) return fail(YAGI_ERR_CONFIG, "decim factor must be greater than 1");             \
        if (m == 0) return fail(YAGI_ERR_CONFIG, "filter delay must be greater than 0");            \
        if (as_ < 0.0f) return fail(YAGI_ERR_CONFIG, "stop-band attenuation must be positive");     \
        std::vector<float> hf;                                                                      \
        YG_TRY(design_kaiser(2 * M * m + 1, 0.5f / (float)M, as_, 0.0f, hf));                       \
        std::vector<C> hc(hf.size());                                                               \
        for (size_t i = 0; i < hf.size(); ++i) hc[i] = to_c(hf[i], (C *)nullptr);                   \
        return yagi_hip_firdecim_##K##_create(M, hc.data(), hc.size(), q);                          \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_destroy(yagi_hip_firdecim_##K q) try {                              \
        delete q;                                                                                   \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_clone(yagi_hip_firdecim_##K q, yagi_hip_firdecim_##K *out) try {    \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(out);                                                                             \
        *out = nullptr;                                                                             \
        YG_TRY(q->w.ensure_dev(q->st));                                                                  \
        auto o = std::make_unique<yagi_hip_firdecim_##K##_s>();                                     \
        o->st = q->st;                                                                              \
        YG_TRY(o->init((size_t)q->M, q->h.data(), q->h.size()));                                    \
        o->scale = q->scale;                                                                        \
        YG_TRY(o->w.clone_from(q->w, q->st));                                                       \
        *out = o.release();                                                                         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_set_stream(yagi_hip_firdecim_##K q, yagi_stream_t s) try {          \
        CHECK_Q(q);                                                                                 \
        if (q->st == to_stream(s)) return YAGI_OK;                                                  \
        YG_HIP(hipStreamSynchronize(q->st));                                                        \
        q->st = to_stream(s);                                                                       \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_reset(yagi_hip_firdecim_##K q) try {                                \
        CHECK_Q(q);                                                                                 \
        return q->w.reset(q->st);                                                                   \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_get_decim_rate(yagi_hip_firdecim_##K q, size_t *M) try {            \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(M);                                                                               \
        *M = (size_t)q->M;                                                                          \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_set_scale(yagi_hip_firdecim_##K q, C s) try {                       \
        CHECK_Q(q);                                                                                 \
        q->scale = s;                                                                               \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_get_scale(yagi_hip_firdecim_##K q, C *s) try {                      \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(s);                                                                               \
        *s = q->scale;                                                                              \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_freqresp(yagi_hip_firdecim_##K q, float fc, yagi_cf32 *H) try {     \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(H);                                                                               \
        *H = cx_mul(taps_freqresponse(q->h, fc), cx_val(q->scale));                                 \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_execute(yagi_hip_firdecim_##K q, const T *x, size_t nx, T *y) try { \
        CHECK_Q(q);                                                                                 \
        if (nx < (size_t)q->M) return fail(YAGI_ERR_CONFIG, "input block too short: need %zu samples, got %zu", (size_t)q->M, nx); \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        YG_TRY(q->w.ensure_host(q->st));                       /* firdecim.rs:179-191: output after the FIRST of the M pushes */ \
        q->w.push(x[0]);                                                                            \
        *y = host_fir_window_dot<T, C>(q->w.host(), (size_t)q->L, q->h.data(), q->scale);           \
        for (int i = 1; i < q->M; ++i) q->w.push(x[i]);                                             \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_execute_block(yagi_hip_firdecim_##K q, const T *x, size_t nx,       \
                                              size_t n, T *y) try {                                 \
        CHECK_Q(q);                                                                                 \
        return firdecim_block_host<KT>(q, x, nx, n, y);                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firdecim_##K##_execute_block_dev(yagi_hip_firdecim_##K q, const T *x, size_t n,    \
                                                  T *y) try {                                       \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, n * (size_t)q->M, y, n);                                                   \
        return q->block_dev(x, n, y);                                                               \
    } catch (...) { return ::yagi::api_exception(); }                                               \
                                                                                                    \
    int yagi_hip_firpfb_##K##_create(size_t nf, const C *h, size_t h_len, yagi_hip_firpfb_##K *q) try { \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (h_len && !h) return fail(YAGI_ERR_CONFIG, "null pointer argument");                     \
        auto o = std::make_unique<yagi_hip_firpfb_##K##_s>();                                       \
        YG_TRY(o->init(nf, h, h_len));                                                              \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_create_kaiser(size_t nf, size_t m, float fc, float as_,               \
                                            yagi_hip_firpfb_##K *q) try {                           \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (nf == 0) return fail(YAGI_ERR_CONFIG, "number of filters must be greater than zero");   \
        if (m == 0) return fail(YAGI_ERR_CONFIG, "filter delay must be greater than 0");            \
        if (fc <= 0.0f || fc > 0.5f)                                                                \
            return fail(YAGI_ERR_CONFIG, "filter cut-off frequency must be in (0,0.5)");            \
        if (as_ < 0.0f) return fail(YAGI_ERR_CONFIG, "filter stop-band suppression must be positive"); \
        std::vector<float> hf;                                                                      \
        YG_TRY(design_kaiser(2 * nf * m + 1, fc / (float)nf, as_, 0.0f, hf));                       \
        std::vector<C> hc(hf.size());                                                               \
        for (size_t i = 0; i < hf.size(); ++i) hc[i] = to_c(hf[i], (C *)nullptr);                   \
        return yagi_hip_firpfb_##K##_create(nf, hc.data(), hc.size(), q);                           \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_create_default(size_t nf, size_t m, yagi_hip_firpfb_##K *q) try {     \
        return yagi_hip_firpfb_##K##_create_kaiser(nf, m, 0.5f, 60.0f, q);                          \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_destroy(yagi_hip_firpfb_##K q) try {                                  \
        delete q;                                                                                   \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_clone(yagi_hip_firpfb_##K q, yagi_hip_firpfb_##K *out) try {          \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(out);                                                                             \
        *out = nullptr;                                                                             \
        YG_TRY(q->w.ensure_dev(q->st));                                                                  \
        auto o = std::make_unique<yagi_hip_firpfb_##K##_s>();                                       \
        o->st = q->st;                                                                              \
        o->nf = q->nf;                                                                              \
        o->Ls = q->Ls;                                                                              \
        o->hb = q->hb;                                                                              \
        o->scale = q->scale;                                                                        \
        YG_TRY(o->taps.alloc(o->hb.size() * sizeof(C)));                                            \
        YG_TRY(upload(o->taps.p, o->hb.data(), o->hb.size() * sizeof(C), o->st));                   \
        YG_TRY(o->w.clone_from(q->w, q->st));                                                       \
        *out = o.release();                                                                         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_set_stream(yagi_hip_firpfb_##K q, yagi_stream_t s) try {              \
        CHECK_Q(q);                                                                                 \
        if (q->st == to_stream(s)) return YAGI_OK;                                                  \
        YG_HIP(hipStreamSynchronize(q->st));                                                        \
        q->st = to_stream(s);                                                                       \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_reset(yagi_hip_firpfb_##K q) try {                                    \
        CHECK_Q(q);                                                                                 \
        return q->w.reset(q->st);                                                                   \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_set_scale(yagi_hip_firpfb_##K q, C s) try {                           \
        CHECK_Q(q);                                                                                 \
        q->scale = s;                                                                               \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_get_scale(yagi_hip_firpfb_##K q, C *s) try {                          \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(s);                                                                               \
        *s = q->scale;                                                                              \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_push(yagi_hip_firpfb_##K q, T x) try {                                \
        CHECK_Q(q);                                                                                 \
        YG_TRY(q->w.ensure_host(q->st));                                                            \
        q->w.push(x);                                                                               \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_write(yagi_hip_firpfb_##K q, const T *x, size_t n) try {              \
        CHECK_Q(q);                                                                                 \
        if (n && !x) return fail(YAGI_ERR_CONFIG, "null pointer argument");                         \
        YG_TRY(q->w.ensure_host(q->st));                                                            \
        for (size_t i = 0; i < n; ++i) q->w.push(x[i]);                                             \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_execute(yagi_hip_firpfb_##K q, size_t i, T *y) try {                  \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(y);                                                                               \
        YG_TRY(q->check_branch(i));                                                                 \
        YG_TRY(q->w.ensure_host(q->st));                                                            \
        *y = host_fir_window_dot<T, C>(q->w.host(), (size_t)q->Ls, q->hb.data() + i * (size_t)q->Ls, q->scale); \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_execute_block(yagi_hip_firpfb_##K q, size_t i, const T *x, size_t nx, \
                                            T *y, size_t ny) try {                                  \
        CHECK_Q(q);                                                                                 \
        return firpfb_block_host<KT>(q, i, x, nx, y, ny);                                           \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_execute_block_dev(yagi_hip_firpfb_##K q, size_t i, const T *x,        \
                                                size_t n, T *y) try {                               \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return q->check_branch(i);                                                      \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, n, y, n);                                                                  \
        return q->block_dev(i, x, n, y);                                                            \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_execute_all_dev(yagi_hip_firpfb_##K q, const T *x, size_t n, T *y) try { \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, n, y, n * (size_t)q->nf);                                                  \
        YG_TRY(q->w.ensure_dev(q->st));                                                                  \
        YG_TRY((launch_firpfb_all<KT>(q->w.dev(), x, q->taps.as<C>(), q->nf, q->Ls, q->scale, y, n, \
                                      q->st)));                                                     \
        return q->w.advance(x, n, q->st);                                                           \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firpfb_##K##_execute_select_dev(yagi_hip_firpfb_##K q, const uint32_t *idx,        \
                                                 const T *x, size_t n, T *y) try {                  \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(idx);                                                                             \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, n, y, n);                                                                  \
        YG_TRY(q->w.ensure_dev(q->st));                                                                  \
        YG_TRY((launch_firpfb_select<KT>(q->w.dev(), x, q->taps.as<C>(), idx, q->nf, q->Ls,         \
                                         q->scale, y, n, q->st)));                                  \
        return q->w.advance(x, n, q->st);                                                           \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    }

YAGI_FIR_IMPL(rrrf, RRRF, float, float)
YAGI_FIR_IMPL(crcf, CRCF, yagi_cf32, float)
YAGI_FIR_IMPL(cccf, CCCF, yagi_cf32, yagi_cf32)

// which block kernel execute_block uses: 0 auto, 1 general direct form, 4 overlap-save fast convolution (<= 2049
// taps); crcf also 2 register-sliding direct form (<= 1024 taps) and 3 MFMA Toeplitz direct form (<= 256 taps)
template <class Q>
static int firfilt_set_kernel(Q *q, int choice, bool crcf) {
    CHECK_Q(q);
    if (choice < 0 || choice > 4 || (!crcf && (choice == 2 || choice == 3)))
        return fail(YAGI_ERR_CONFIG, "unknown kernel choice %d", choice);
    if (choice == 4 && q->L > 2049) return fail(YAGI_ERR_CONFIG, "fast convolution needs <= 2049 taps");
    q->kernel_choice = choice;
    return YAGI_OK;
}
extern "C" int yagi_hip_firfilt_rrrf_set_kernel(yagi_hip_firfilt_rrrf q, int choice) { return firfilt_set_kernel(q, choice, false); }
extern "C" int yagi_hip_firfilt_crcf_set_kernel(yagi_hip_firfilt_crcf q, int choice) { return firfilt_set_kernel(q, choice, true); }
extern "C" int yagi_hip_firfilt_cccf_set_kernel(yagi_hip_firfilt_cccf q, int choice) { return firfilt_set_kernel(q, choice, false); }

// ---- Fft --------------------------------------------------------------------------------------------
struct yagi_hip_fft_s : FftPlan {};

extern "C" {

int yagi_hip_fft_create(size_t n, int direction, yagi_hip_fft *plan) try {
    CHECK_PTR(plan);
    *plan = nullptr;
    auto p = std::make_unique<yagi_hip_fft_s>();
    YG_TRY(fft_plan_init(*p, n, direction));
    *plan = p.release();
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_fft_destroy(yagi_hip_fft plan) try { delete plan; return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_fft_clone(yagi_hip_fft plan, yagi_hip_fft *out) try {
    CHECK_Q(plan);
    return yagi_hip_fft_create((size_t)plan->d.n, plan->d.dir, out);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_fft_len(yagi_hip_fft plan, size_t *n) try {
    CHECK_Q(plan);
    CHECK_PTR(n);
    *n = (size_t)plan->d.n;
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_fft_run_batch_dev(yagi_hip_fft plan, const yagi_cf32 *in, yagi_cf32 *out, size_t batch,
                               yagi_stream_t s) try {
    CHECK_Q(plan);
    if (batch == 0) return YAGI_OK;
    CHECK_PTR(in);
    CHECK_PTR(out);
    return launch_fft_batch(plan->d, in, out, batch, to_stream(s));
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_fft_run(yagi_hip_fft plan, const yagi_cf32 *input, size_t n_in, yagi_cf32 *output, size_t n_out) try {
    CHECK_Q(plan);
    const size_t n = (size_t)plan->d.n;
    // the reference panics on a length mismatch (copy_from_slice, fft/mod.rs:46)
    if (n_in != n || n_out != n) return fail(YAGI_ERR_CONFIG, "fft buffers must hold exactly %zu samples", n);
    CHECK_PTR(input);
    CHECK_PTR(output);
    YG_TRY(plan->ws.x.ensure(n * sizeof(cf32)));
    YG_TRY(plan->ws.y.ensure(n * sizeof(cf32)));
    YG_TRY(upload(plan->ws.x.p, input, n * sizeof(cf32), nullptr));
    YG_TRY(launch_fft_batch(plan->d, plan->ws.x.as<cf32>(), plan->ws.y.as<cf32>(), 1, nullptr));
    return download(output, plan->ws.y.p, n * sizeof(cf32), nullptr);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_fft_shift_dev(yagi_cf32 *buf, size_t n, size_t batch, yagi_stream_t s) try {
    if (n == 0 || batch == 0) return YAGI_OK;
    CHECK_PTR(buf);
    YG_TRY(require_device());
    return launch_fft_shift(buf, n, batch, to_stream(s));
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_fft_shift(yagi_cf32 *buf, size_t n) try {
    if (n < 2) return YAGI_OK;
    CHECK_PTR(buf);
    YG_TRY(require_device());
    DevBuf d;
    YG_TRY(d.alloc(n * sizeof(cf32)));
    YG_TRY(upload(d.p, buf, n * sizeof(cf32), nullptr));
    YG_TRY(launch_fft_shift(d.as<cf32>(), n, 1, nullptr));
    return download(buf, d.p, n * sizeof(cf32), nullptr);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_fft_run_oneshot(const yagi_cf32 *input, yagi_cf32 *output, size_t n, int direction) try {
    yagi_hip_fft p = nullptr;
    YG_TRY(yagi_hip_fft_create(n, direction, &p));
    int rc = yagi_hip_fft_run(p, input, n, output, n);
    yagi_hip_fft_destroy(p);
    return rc;
} catch (...) { return ::yagi::api_exception(); }

}  // extern "C"

// ---- FirInterpolationFilter (src/filter/fir/firinterp.rs): a FirPfb bank run over every branch -------
namespace yagi {

template <class K>
struct FirInterp {
    using T = typename K::T;
    using C = typename K::C;
    FirPfb<K> bank;
    int interp = 0, hs = 0;

    int init(size_t m, const C *h, size_t h_len) {
        if (m < 2) return fail(YAGI_ERR_CONFIG, "interp factor must be greater than 1");
        if (h_len < m) return fail(YAGI_ERR_CONFIG, "filter length cannot be less than interp factor");
        size_t sub = 0;
        while (m * sub < h_len) ++sub;                       // firinterp.rs:44-47
        std::vector<C> hp(m * sub, C{});
        for (size_t i = 0; i < h_len; ++i) hp[i] = h[i];
        YG_TRY(bank.init(m, hp.data(), hp.size()));
        interp = (int)m;
        hs = (int)sub;
        return YAGI_OK;
    }
    // n inputs -> n*interp outputs (device pointers)
    int block_dev(const T *x, size_t n, T *y) {
        YG_TRY(bank.w.ensure_dev(bank.st));
        if (n == 0) return YAGI_OK;
        YG_TRY((launch_firpfb_all<K>(bank.w.dev(), x, bank.taps.template as<C>(), bank.nf, bank.Ls,
                                     bank.scale, y, n, bank.st, bank.w.next())));
        bank.w.flip();                              // the kernel's last workgroup wrote the next window
        return YAGI_OK;
    }
    int block_host(const T *x, size_t n, T *y) {
        if (n == 0) return YAGI_OK;
        YG_TRY(bank.ws.x.ensure(n * sizeof(T)));
        YG_TRY(bank.ws.y.ensure(n * (size_t)interp * sizeof(T)));
        YG_TRY(upload(bank.ws.x.p, x, n * sizeof(T), bank.st));
        YG_TRY(block_dev(bank.ws.x.template as<T>(), n, bank.ws.y.template as<T>()));
        return download(y, bank.ws.y.p, n * (size_t)interp * sizeof(T), bank.st);
    }
};

}  // namespace yagi

#define YAGI_FIRINTERP_IMPL(K, KT, T, C)                                                            \
    struct yagi_hip_firinterp_##K##_s : FirInterp<KT> {};                                           \
    extern "C" {                                                                                    \
    int yagi_hip_firinterp_##K##_create(size_t interp, const C *h, size_t h_len,                    \
                                        yagi_hip_firinterp_##K *q) try {                            \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (h_len && !h) return fail(YAGI_ERR_CONFIG, "null pointer argument");                     \
        auto o = std::make_unique<yagi_hip_firinterp_##K##_s>();                                    \
        YG_TRY(o->init(interp, h, h_len));                                                          \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_create_kaiser(size_t interp, size_t m, float as_,                  \
                                               yagi_hip_firinterp_##K *q) try {                     \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (interp < 2) return fail(YAGI_ERR_CONFIG, "interp factor must be greater than 1");       \
        if (m == 0) return fail(YAGI_ERR_CONFIG, "filter delay must be greater than 0");            \
        if (as_ < 0.0f) return fail(YAGI_ERR_CONFIG, "stop-band attenuation must be positive");     \
        std::vector<float> hf;                                                                      \
        YG_TRY(design_kaiser(2 * interp * m + 1, 0.5f / (float)interp, as_, 0.0f, hf));             \
        std::vector<C> hc(hf.size());                                                               \
        for (size_t i = 0; i < hf.size(); ++i) hc[i] = to_c(hf[i], (C *)nullptr);                   \
        return yagi_hip_firinterp_##K##_create(interp, hc.data(), hc.size() - 1, q);                \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_create_linear(size_t interp, yagi_hip_firinterp_##K *q) try {      \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (interp < 1) return fail(YAGI_ERR_CONFIG, "interp factor must be greater than 1");       \
        std::vector<C> hc(2 * interp);                                                              \
        for (size_t i = 0; i < interp; ++i) {                                                       \
            hc[i] = to_c((float)i / (float)interp, (C *)nullptr);                                   \
            hc[interp + i] = to_c(1.0f - (float)i / (float)interp, (C *)nullptr);                   \
        }                                                                                           \
        return yagi_hip_firinterp_##K##_create(interp, hc.data(), hc.size(), q);                    \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_create_window(size_t interp, size_t m, yagi_hip_firinterp_##K *q) try { \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (interp < 1) return fail(YAGI_ERR_CONFIG, "interp factor must be greater than 1");       \
        if (m < 1) return fail(YAGI_ERR_CONFIG, "filter semi-length must be greater than 0");       \
        const size_t hl = 2 * m * interp;                                                           \
        std::vector<C> hc(hl);                                                                      \
        for (size_t i = 0; i < hl; ++i) {                                                           \
            const float sv = std::sin(3.14159265358979323846f * (float)i / (float)(2 * m * interp));\
            hc[i] = to_c(sv * sv, (C *)nullptr);                                                    \
        }                                                                                           \
        return yagi_hip_firinterp_##K##_create(interp, hc.data(), hl, q);                           \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_destroy(yagi_hip_firinterp_##K q) try {                            \
        delete q;                                                                                   \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_clone(yagi_hip_firinterp_##K q, yagi_hip_firinterp_##K *out) try { \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(out);                                                                             \
        *out = nullptr;                                                                             \
        YG_TRY(q->bank.w.ensure_dev(q->bank.st));                                                        \
        auto o = std::make_unique<yagi_hip_firinterp_##K##_s>();                                    \
        o->interp = q->interp;                                                                      \
        o->hs = q->hs;                                                                              \
        o->bank.st = q->bank.st;                                                                    \
        o->bank.nf = q->bank.nf;                                                                    \
        o->bank.Ls = q->bank.Ls;                                                                    \
        o->bank.hb = q->bank.hb;                                                                    \
        o->bank.scale = q->bank.scale;                                                              \
        YG_TRY(o->bank.taps.alloc(o->bank.hb.size() * sizeof(C)));                                  \
        YG_TRY(upload(o->bank.taps.p, o->bank.hb.data(), o->bank.hb.size() * sizeof(C), o->bank.st)); \
        YG_TRY(o->bank.w.clone_from(q->bank.w, q->bank.st));                                        \
        *out = o.release();                                                                         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_set_stream(yagi_hip_firinterp_##K q, yagi_stream_t s) try {        \
        CHECK_Q(q);                                                                                 \
        if (q->bank.st == to_stream(s)) return YAGI_OK;                                             \
        YG_HIP(hipStreamSynchronize(q->bank.st));                                                   \
        q->bank.st = to_stream(s);                                                                  \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_reset(yagi_hip_firinterp_##K q) try {                              \
        CHECK_Q(q);                                                                                 \
        return q->bank.w.reset(q->bank.st);                                                         \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_get_interp_rate(yagi_hip_firinterp_##K q, size_t *interp) try {    \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(interp);                                                                          \
        *interp = (size_t)q->interp;                                                                \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_get_sub_len(yagi_hip_firinterp_##K q, size_t *hs) try {            \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(hs);                                                                              \
        *hs = (size_t)q->hs;                                                                        \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_set_scale(yagi_hip_firinterp_##K q, C scale) try {                 \
        CHECK_Q(q);                                                                                 \
        q->bank.scale = scale;                                                                      \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_get_scale(yagi_hip_firinterp_##K q, C *scale) try {                \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(scale);                                                                           \
        *scale = q->bank.scale;                                                                     \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_execute(yagi_hip_firinterp_##K q, T x, T *y, size_t ny) try {      \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(y);                                                                               \
        if (ny < (size_t)q->interp) return fail(YAGI_ERR_CONFIG, "output must hold interp samples");\
        return q->block_host(&x, 1, y);                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_execute_block(yagi_hip_firinterp_##K q, const T *x, size_t nx,     \
                                               T *y, size_t ny) try {                               \
        CHECK_Q(q);                                                                                 \
        if (nx == 0) return YAGI_OK;                                                                \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        if (ny < nx * (size_t)q->interp)                                                            \
            return fail(YAGI_ERR_CONFIG, "output must hold n*interp samples");                      \
        return q->block_host(x, nx, y);                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_execute_block_dev(yagi_hip_firinterp_##K q, const T *x, size_t n,  \
                                                   T *y) try {                                      \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, n, y, n * (size_t)q->bank.nf);                                             \
        return q->block_dev(x, n, y);                                                               \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_firinterp_##K##_flush(yagi_hip_firinterp_##K q, T *y, size_t ny) try {             \
        T zero{};                                                                                   \
        return yagi_hip_firinterp_##K##_execute(q, zero, y, ny);                                    \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    }

YAGI_FIRINTERP_IMPL(rrrf, RRRF, float, float)
YAGI_FIRINTERP_IMPL(crcf, CRCF, yagi_cf32, float)
YAGI_FIRINTERP_IMPL(cccf, CCCF, yagi_cf32, yagi_cf32)

// ---- Rresamp (src/filter/resampler/rresamp.rs) ----------------------------------------------------------
namespace yagi {

static size_t gcd_sz(size_t a, size_t b) {
    while (b) { const size_t t = a % b; a = b; b = t; }
    return a;
}

template <class K>
struct RresampObj {
    using T = typename K::T;
    using C = typename K::C;
    FirPfb<K> bank;                      // P branches of 2m taps (rresamp.rs:40)
    int P = 0, Q = 0, m = 0, block_len = 1;

    int init(size_t interp, size_t decim, size_t m_, const C *h, size_t h_len) {      // :28-57
        if (interp == 0) return fail(YAGI_ERR_CONFIG, "interpolation rate must be greater than zero");
        if (decim == 0) return fail(YAGI_ERR_CONFIG, "decimation rate must be greater than zero");
        if (m_ == 0) return fail(YAGI_ERR_CONFIG, "filter semi-length must be greater than zero");
        if (interp > (size_t)1 << 15 || decim > (size_t)1 << 15 || m_ > (size_t)1 << 15)
            return fail(YAGI_ERR_CONFIG, "rresamp: rate or filter too large");
        if (h_len < 2 * interp * m_) return fail(YAGI_ERR_CONFIG, "rresamp: need 2*interp*m filter coefficients");
        YG_TRY(bank.init(interp, h, 2 * interp * m_));
        P = (int)interp; Q = (int)decim; m = (int)m_;
        block_len = 1;
        return YAGI_OK;
    }
    // nblocks primitive blocks: nblocks*Q inputs -> nblocks*P outputs (device pointers); :162-183
    int blocks_dev(const T *x, size_t nblocks, T *y) {
        YG_TRY(bank.w.ensure_dev(bank.st));
        if (nblocks == 0) return YAGI_OK;
        YG_TRY((launch_rresamp<K>(bank.w.dev(), x, bank.taps.template as<C>(), P, Q, bank.Ls, bank.scale, y,
                                  nblocks, bank.st, bank.w.next())));
        bank.w.flip();                              // the kernel's last workgroup wrote the next window
        return YAGI_OK;
    }
    int blocks_host(const T *x, size_t nblocks, T *y) {
        if (nblocks == 0) return YAGI_OK;
        YG_TRY(bank.ws.x.ensure(nblocks * (size_t)Q * sizeof(T)));
        YG_TRY(bank.ws.y.ensure(nblocks * (size_t)P * sizeof(T)));
        YG_TRY(upload(bank.ws.x.p, x, nblocks * (size_t)Q * sizeof(T), bank.st));
        YG_TRY(blocks_dev(bank.ws.x.template as<T>(), nblocks, bank.ws.y.template as<T>()));
        return download(y, bank.ws.y.p, nblocks * (size_t)P * sizeof(T), bank.st);
    }
};

}  // namespace yagi

#define YAGI_RRESAMP_IMPL(K, KT, T, C)                                                              \
    struct yagi_hip_rresamp_##K##_s : RresampObj<KT> {};                                            \
    extern "C" {                                                                                    \
    int yagi_hip_rresamp_##K##_create(size_t interp, size_t decim, size_t m, const C *h,            \
                                      size_t h_len, yagi_hip_rresamp_##K *q) try {                  \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (h_len && !h) return fail(YAGI_ERR_CONFIG, "null pointer argument");                     \
        auto o = std::make_unique<yagi_hip_rresamp_##K##_s>();                                      \
        YG_TRY(o->init(interp, decim, m, h, h_len));                                                \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_create_kaiser(size_t interp, size_t decim, size_t m, float bw,       \
                                             float as_, yagi_hip_rresamp_##K *q) try {              \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (interp == 0 || decim == 0) return fail(YAGI_ERR_CONFIG, "gcd: arguments must be greater than zero"); \
        const size_t g = gcd_sz(interp, decim);                          /* rresamp.rs:60-62 */     \
        interp /= g;                                                                                \
        decim /= g;                                                                                 \
        if (bw < 0.0f) bw = interp > decim ? 0.5f : 0.5f * (float)interp / (float)decim;            \
        else if (bw > 0.5f) return fail(YAGI_ERR_CONFIG, "invalid bandwidth (%g), must be less than 0.5", (double)bw); \
        if (m == 0) return fail(YAGI_ERR_CONFIG, "filter semi-length must be greater than zero");   \
        std::vector<float> hf;                                                                      \
        YG_TRY(design_kaiser(2 * interp * m + 1, bw / (float)interp, as_, 0.0f, hf));               \
        std::vector<C> hc(hf.size());                                                               \
        for (size_t i = 0; i < hf.size(); ++i) hc[i] = to_c(hf[i], (C *)nullptr);                   \
        YG_TRY(yagi_hip_rresamp_##K##_create(interp, decim, m, hc.data(), hc.size(), q));           \
        (*q)->bank.scale = to_c(2.0f * bw * std::sqrt((float)decim / (float)interp), (C *)nullptr); \
        (*q)->block_len = (int)g;                                                                   \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_create_default(size_t interp, size_t decim, yagi_hip_rresamp_##K *q) try { \
        return yagi_hip_rresamp_##K##_create_kaiser(interp, decim, 12, 0.5f, 60.0f, q);  /* :99-104 */ \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_destroy(yagi_hip_rresamp_##K q) try {                                \
        delete q;                                                                                   \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_clone(yagi_hip_rresamp_##K q, yagi_hip_rresamp_##K *out) try {  /* derive(Clone) rresamp.rs:8 */ \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(out);                                                                             \
        *out = nullptr;                                                                             \
        YG_TRY(q->bank.w.ensure_dev(q->bank.st));                                                        \
        auto o = std::make_unique<yagi_hip_rresamp_##K##_s>();                                      \
        o->bank.st = q->bank.st;                                                                    \
        o->bank.nf = q->bank.nf;                                                                    \
        o->bank.Ls = q->bank.Ls;                                                                    \
        o->bank.hb = q->bank.hb;                                                                    \
        o->bank.scale = q->bank.scale;                                                              \
        YG_TRY(o->bank.taps.alloc(o->bank.hb.size() * sizeof(C)));                                  \
        YG_TRY(upload(o->bank.taps.p, o->bank.hb.data(), o->bank.hb.size() * sizeof(C), o->bank.st)); \
        YG_TRY(o->bank.w.clone_from(q->bank.w, q->bank.st));                                        \
        o->P = q->P; o->Q = q->Q; o->m = q->m; o->block_len = q->block_len;                         \
        *out = o.release();                                                                         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_set_stream(yagi_hip_rresamp_##K q, yagi_stream_t s) try {            \
        CHECK_Q(q);                                                                                 \
        if (q->bank.st == to_stream(s)) return YAGI_OK;                                             \
        YG_HIP(hipStreamSynchronize(q->bank.st));                                                   \
        q->bank.st = to_stream(s);                                                                  \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_reset(yagi_hip_rresamp_##K q) try {                                  \
        CHECK_Q(q);                                                                                 \
        return q->bank.w.reset(q->bank.st);                                                         \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_set_scale(yagi_hip_rresamp_##K q, C scale) try {                     \
        CHECK_Q(q);                                                                                 \
        q->bank.scale = scale;                                                                      \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_get_scale(yagi_hip_rresamp_##K q, C *scale) try {                    \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(scale);                                                                           \
        *scale = q->bank.scale;                                                                     \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_get_params(yagi_hip_rresamp_##K q, size_t *interp, size_t *decim,    \
                                          size_t *m, size_t *block_len) try {                       \
        CHECK_Q(q);                                                                                 \
        if (interp) *interp = (size_t)q->P;                                                         \
        if (decim) *decim = (size_t)q->Q;                                                           \
        if (m) *m = (size_t)q->m;                                                                   \
        if (block_len) *block_len = (size_t)q->block_len;                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_write(yagi_hip_rresamp_##K q, const T *x, size_t n) try {            \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        YG_TRY(q->bank.w.ensure_host(q->bank.st));                                                  \
        for (size_t i = 0; i < n; ++i) q->bank.w.push(x[i]);                                        \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_execute(yagi_hip_rresamp_##K q, const T *x, size_t nx, T *y,         \
                                       size_t ny) try {                                             \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        const size_t bl = (size_t)q->block_len;                                                     \
        if (nx < bl * (size_t)q->Q) return fail(YAGI_ERR_RANGE, "input must hold Q*block_len samples"); \
        if (ny < bl * (size_t)q->P) return fail(YAGI_ERR_RANGE, "output must hold P*block_len samples"); \
        return q->blocks_host(x, bl, y);                                                            \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_execute_block(yagi_hip_rresamp_##K q, const T *x, size_t nx,         \
                                             size_t n, T *y, size_t ny) try {                       \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        const size_t nb = n * (size_t)q->block_len;                                                 \
        if (nx < nb * (size_t)q->Q) return fail(YAGI_ERR_RANGE, "input must hold n*Q*block_len samples"); \
        if (ny < nb * (size_t)q->P) return fail(YAGI_ERR_RANGE, "output must hold n*P*block_len samples"); \
        return q->blocks_host(x, nb, y);                                                            \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_rresamp_##K##_execute_block_dev(yagi_hip_rresamp_##K q, const T *x, size_t n,      \
                                                 T *y) try {                                        \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, n * (size_t)q->block_len * (size_t)q->Q, y, n * (size_t)q->block_len * (size_t)q->P); \
        return q->blocks_dev(x, n * (size_t)q->block_len, y);                                       \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    }

YAGI_RRESAMP_IMPL(rrrf, RRRF, float, float)
YAGI_RRESAMP_IMPL(crcf, CRCF, yagi_cf32, float)
YAGI_RRESAMP_IMPL(cccf, CCCF, yagi_cf32, yagi_cf32)

// ---- Spgram (src/fft/spgram.rs) -----------------------------------------------------------------------
namespace yagi {

template <class T>
struct SpgramObj {
    hipStream_t st = nullptr;
    int nfft = 0, wtype = 0, wlen = 0, delay = 0;
    float alpha = 1.0f, gamma = 1.0f;
    bool accumulate = true;
    DevWindow<T> buf;
    DevBuf w, psd, out, tbuf, fbuf, part;
    FftPlan plan;
    Workspace ws;
    std::vector<T> queue;                // samples push()ed since the last flush
    size_t sample_timer = 0;
    uint64_t num_samples = 0, num_samples_total = 0, num_transforms = 0, num_transforms_total = 0;
    float frequency = 0.0f, sample_rate = -1.0f;

    int init(size_t nfft_, int wtype_, size_t wlen_, size_t delay_) {
        if (nfft_ < 2) return fail(YAGI_ERR_CONFIG, "fft size must be at least 2");
        if (wlen_ > nfft_) return fail(YAGI_ERR_CONFIG, "window size cannot exceed fft size");
        if (wlen_ == 0) return fail(YAGI_ERR_CONFIG, "window size must be greater than zero");
        if (wtype_ == YAGI_WINDOW_KAISER && wlen_ % 2 != 0)              // spgram.rs:60-62 (sic)
            return fail(YAGI_ERR_CONFIG, "KBD window length must be even");
        if (delay_ == 0) return fail(YAGI_ERR_CONFIG, "delay must be greater than 0");
        if (wtype_ < 1 || wtype_ > 9) return fail(YAGI_ERR_CONFIG, "unknown window type");
        if (nfft_ > ((size_t)1 << 20)) return fail(YAGI_ERR_CONFIG, "fft size %zu not supported (max %d)", nfft_, 1 << 20);
        YG_TRY(require_device());
        nfft = (int)nfft_; wtype = wtype_; wlen = (int)wlen_; delay = (int)delay_;
        // taper (spgram.rs:93-119): window, then normalise to unit energy
        std::vector<float> wv(wlen_);
        float g = 0.0f;
        for (size_t i = 0; i < wlen_; ++i) {
            float arg = 0.0f;
            switch (wtype) {
                case YAGI_WINDOW_KAISER: arg = 10.0f; break;
                case YAGI_WINDOW_KBD: arg = 3.0f; break;
                case YAGI_WINDOW_TRIANGULAR: arg = (float)wlen_; break;
                case YAGI_WINDOW_RCOSTAPER: arg = (float)(wlen_ / 3); break;
                default: break;
            }
            YG_TRY(window_value(wtype, i, wlen_, arg, &wv[i]));
            g += wv[i] * wv[i];
        }
        g = 1.0f / std::sqrt(g);
        for (auto &v : wv) v = g * v;
        YG_TRY(w.alloc(wlen_ * sizeof(float)));
        YG_TRY(upload(w.p, wv.data(), wlen_ * sizeof(float), st));
        YG_TRY(psd.alloc(nfft_ * sizeof(float)));
        YG_TRY(out.alloc(nfft_ * sizeof(float)));
        YG_TRY(fft_plan_init(plan, nfft_, YAGI_FFT_FORWARD));
        YG_TRY(buf.init(wlen, st));
        set_alpha_unchecked(-1.0f);
        return reset();
    }
    void set_alpha_unchecked(float a) {
        accumulate = (a == -1.0f);
        if (accumulate) { alpha = 1.0f; gamma = 1.0f; } else { alpha = a; gamma = 1.0f - a; }
    }
    int clear() {                                                    // :135-147
        queue.clear();
        sample_timer = (size_t)delay;
        num_transforms = 0;
        num_samples = 0;
        YG_HIP(hipMemsetAsync(psd.p, 0, (size_t)nfft * sizeof(float), st));
        return YAGI_OK;
    }
    int reset() {                                                    // :150-155
        YG_TRY(clear());
        YG_TRY(buf.reset(st));
        num_samples_total = 0;
        num_transforms_total = 0;
        return YAGI_OK;
    }
    // transforms whose newest sample is X[first + f*delay], f < nframes (X = window ++ x)
    // x_len = samples readable at x (the write() block)
    int run_frames(const T *x, long long first, size_t nframes, size_t x_len = 0) {
        if (spgram_fused_supported(nfft)) {       // fused taper -> FFT -> |X|^2 -> accumulate (spgram_kernels.hip)
            const size_t big = (size_t)1 << 20;   // transforms per launch
            for (size_t f0 = 0; f0 < nframes; f0 += big) {
                const size_t nf = (nframes - f0) < big ? (nframes - f0) : big;
                YG_TRY(part.ensure(spgram_fused_scratch_floats(nfft, nf) * sizeof(float)));
                YG_TRY(launch_spgram_fused<T>(nfft, buf.dev(), x, x_len, w.as<float>(), wlen, first + (long long)f0 * delay, delay, nf,
                                                  alpha, gamma, num_transforms == 0, plan.d.tw, psd.as<float>(),
                                                  part.as<float>(), st));
                num_transforms += nf;
                num_transforms_total += nf;
            }
            return YAGI_OK;
        }
        size_t chunk = ((size_t)1 << 25) / (size_t)nfft;          // transforms per batch: 2 x 256 MiB of frames at most
        chunk = chunk < 1 ? 1 : (chunk > 8192 ? 8192 : chunk);
        for (size_t f0 = 0; f0 < nframes; f0 += chunk) {
            const size_t nf = (nframes - f0) < chunk ? (nframes - f0) : chunk;
            YG_TRY(tbuf.ensure(nf * (size_t)nfft * sizeof(cf32)));
            YG_TRY(fbuf.ensure(nf * (size_t)nfft * sizeof(cf32)));
            YG_TRY(launch_spgram_frames<T>(buf.dev(), x, w.as<float>(), wlen, nfft, first + (long long)f0 * delay, delay,
                                           nf, tbuf.as<cf32>(), st));
            YG_TRY(launch_fft_batch(plan.d, tbuf.as<cf32>(), fbuf.as<cf32>(), nf, st));
            YG_TRY(part.ensure(spgram_accum_scratch_floats(nfft, nf) * sizeof(float)));
            YG_TRY(launch_spgram_accum(fbuf.as<cf32>(), nfft, nf, alpha, gamma, num_transforms == 0, psd.as<float>(),
                                       part.as<float>(), st));
            num_transforms += nf;
            num_transforms_total += nf;
        }
        return YAGI_OK;
    }
    int write_dev(const T *x, size_t n) {                            // push() x n, :237-259
        if (n == 0) return YAGI_OK;
        const size_t t = sample_timer;                               // 1..delay pushes until the next transform
        size_t nframes = 0;
        if (n >= t) nframes = (n - t) / (size_t)delay + 1;
        if (nframes) YG_TRY(run_frames(x, (long long)t - 1, nframes, n));
        sample_timer = (n < t) ? t - n : (size_t)delay - (n - t) % (size_t)delay;
        num_samples += n;
        num_samples_total += n;
        return buf.advance(x, n, st);
    }
    int write_host(const T *x, size_t n) {
        if (n == 0) return YAGI_OK;
        YG_TRY(ws.x.ensure(n * sizeof(T)));
        YG_TRY(upload(ws.x.p, x, n * sizeof(T), st));
        return write_dev(ws.x.as<T>(), n);
    }
    int flush() {
        if (queue.empty()) return YAGI_OK;
        std::vector<T> q2;
        q2.swap(queue);
        return write_host(q2.data(), q2.size());
    }
    int get(float *o, size_t n, bool db) {                           // :292-316
        if (n < (size_t)nfft) return fail(YAGI_ERR_CONFIG, "psd buffer must hold nfft values");
        YG_TRY(flush());
        const uint64_t nt = num_transforms ? num_transforms : 1;
        const float scale = accumulate ? 1.0f / (float)nt : 0.0f;    // the reference's 0.0 when not accumulating
        YG_TRY(launch_spgram_psd(psd.as<float>(), nfft, scale, db, out.as<float>(), st));
        return download(o, out.p, (size_t)nfft * sizeof(float), st);
    }
};

}  // namespace yagi

#define YAGI_SPGRAM_IMPL(K, T)                                                                      \
    struct yagi_hip_spgram##K##_s : SpgramObj<T> {};                                                \
    extern "C" {                                                                                    \
    int yagi_hip_spgram##K##_create(size_t nfft, int wtype, size_t window_len, size_t delay,        \
                                    yagi_hip_spgram##K *q) try {                                    \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        auto o = std::make_unique<yagi_hip_spgram##K##_s>();                                        \
        YG_TRY(o->init(nfft, wtype, window_len, delay));                                            \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_create_default(size_t nfft, yagi_hip_spgram##K *q) try {               \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (nfft < 2) return fail(YAGI_ERR_CONFIG, "fft size must be at least 2");                  \
        return yagi_hip_spgram##K##_create(nfft, YAGI_WINDOW_KAISER, nfft / 2, nfft / 4, q);        \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_destroy(yagi_hip_spgram##K q) try {                                    \
        delete q;                                                                                   \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_set_stream(yagi_hip_spgram##K q, yagi_stream_t s) try {                \
        CHECK_Q(q);                                                                                 \
        if (q->st == to_stream(s)) return YAGI_OK;                                                  \
        YG_HIP(hipStreamSynchronize(q->st));                                                        \
        q->st = to_stream(s);                                                                       \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_clear(yagi_hip_spgram##K q) try {                                      \
        CHECK_Q(q);                                                                                 \
        YG_TRY(q->flush());                                                                         \
        return q->clear();                                                                          \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_reset(yagi_hip_spgram##K q) try {                                      \
        CHECK_Q(q);                                                                                 \
        return q->reset();                                                                          \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_set_alpha(yagi_hip_spgram##K q, float alpha) try {                     \
        CHECK_Q(q);                                                                                 \
        if (alpha != -1.0f && (alpha < 0.0f || alpha > 1.0f))                                       \
            return fail(YAGI_ERR_CONFIG, "alpha must be in {-1,[0,1]}");                            \
        YG_TRY(q->flush());                                                                         \
        q->set_alpha_unchecked(alpha);                                                              \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_get_alpha(yagi_hip_spgram##K q, float *alpha) try {                    \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(alpha);                                                                           \
        *alpha = q->alpha;                                                                          \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_set_freq(yagi_hip_spgram##K q, float freq) try {                       \
        CHECK_Q(q);                                                                                 \
        q->frequency = freq;                                                                        \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_set_rate(yagi_hip_spgram##K q, float rate) try {                       \
        CHECK_Q(q);                                                                                 \
        if (rate <= 0.0f) return fail(YAGI_ERR_CONFIG, "sample rate must be greater than zero");    \
        q->sample_rate = rate;                                                                      \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_get_params(yagi_hip_spgram##K q, size_t *nfft, size_t *window_len,     \
                                        size_t *delay, int *wtype) try {                            \
        CHECK_Q(q);                                                                                 \
        if (nfft) *nfft = (size_t)q->nfft;                                                          \
        if (window_len) *window_len = (size_t)q->wlen;                                              \
        if (delay) *delay = (size_t)q->delay;                                                       \
        if (wtype) *wtype = q->wtype;                                                               \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_get_counters(yagi_hip_spgram##K q, uint64_t *ns, uint64_t *nst,        \
                                          uint64_t *nt, uint64_t *ntt) try {                        \
        CHECK_Q(q);                                                                                 \
        YG_TRY(q->flush());                                                                         \
        if (ns) *ns = q->num_samples;                                                               \
        if (nst) *nst = q->num_samples_total;                                                       \
        if (nt) *nt = q->num_transforms;                                                            \
        if (ntt) *ntt = q->num_transforms_total;                                                    \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_push(yagi_hip_spgram##K q, T x) try {                                  \
        CHECK_Q(q);                                                                                 \
        q->queue.push_back(x);                                                                      \
        if (q->queue.size() >= (size_t)1 << 20) return q->flush();                                  \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_write(yagi_hip_spgram##K q, const T *x, size_t n) try {                \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        YG_TRY(q->flush());                                                                         \
        return q->write_host(x, n);                                                                 \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_write_dev(yagi_hip_spgram##K q, const T *x, size_t n) try {            \
        CHECK_Q(q);                                                                                 \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        YG_TRY(q->flush());                                                                         \
        return q->write_dev(x, n);                                                                  \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_get_psd_mag(yagi_hip_spgram##K q, float *psd, size_t n) try {          \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(psd);                                                                             \
        return q->get(psd, n, false);                                                               \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_get_psd(yagi_hip_spgram##K q, float *psd, size_t n) try {              \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(psd);                                                                             \
        return q->get(psd, n, true);                                                                \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_spgram##K##_estimate_psd(size_t nfft, const T *x, size_t n, float *psd) try {      \
        CHECK_PTR(psd);                                                                             \
        if (n && !x) return fail(YAGI_ERR_CONFIG, "null pointer argument");                         \
        yagi_hip_spgram##K q = nullptr;                                                             \
        YG_TRY(yagi_hip_spgram##K##_create_default(nfft, &q));                                      \
        std::unique_ptr<yagi_hip_spgram##K##_s> guard(q);                                           \
        YG_TRY(q->write_host(x, n));                                                                \
        if (q->num_transforms == 0) YG_TRY(q->run_frames(nullptr, -1, 1));   /* q.step() :325-327 */ \
        return q->get(psd, nfft, true);                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    }

YAGI_SPGRAM_IMPL(cf, yagi_cf32)
YAGI_SPGRAM_IMPL(f, float)

// ---- FftFilt ---------------------------------------------------------------------------------------
namespace yagi {

static cf32 cx_of(float v) { return cf32{v, 0.0f}; }
static cf32 cx_of(cf32 v) { return v; }
static float div_scalar(float s, float d) { return s / d; }
static cf32 div_scalar(cf32 s, float d) { return cf32{s.re / d, s.im / d}; }
static float mul_scalar(float s, float d) { return s * d; }
static cf32 mul_scalar(cf32 s, float d) { return cf32{s.re * d, s.im * d}; }

template <class K>
struct FftFiltObj {
    using T = typename K::T;
    using C = typename K::C;
    hipStream_t st = nullptr;
    std::vector<C> h;
    int n = 0;
    C scale = one_of<C>();          // stored divided by 2n like the reference (fftfilt.rs:95-97)
    FftPlan fwd, bwd;
    DevBuf hfreq;                   // FFT{[h;0]}, 2n points
    DevBuf w[2];                    // overlap tail, n points, ping-pong
    int cur = 0;
    DevBuf tbuf, fbuf;              // [nblocks][2n] time / frequency buffers
    Workspace ws;
    // h_len <= 2049: the blocks of one call are contiguous in the stream and what FftFilt computes is the
    // stream's linear convolution with h, whatever the block length -- so the call goes to the one-launch
    // overlap-save kernel (4096-point transforms chained in registers) and the carried state is the last h_len
    // input samples instead of the overlap-add tail.  Longer filters keep the reference's five stages.
    bool use_conv = false;
    FirFilt<K> fir;

    int init(const C *hh, size_t h_len, size_t nn) {
        if (h_len == 0) return fail(YAGI_ERR_CONFIG, "filter length must be greater than zero");
        if (nn < h_len - 1) return fail(YAGI_ERR_CONFIG, "block length must be greater than h_len-1 (%zu)", h_len - 1);
        if (nn == 0) return fail(YAGI_ERR_CONFIG, "block length must be greater than zero");
        if (2 * nn > ((size_t)1 << 22)) return fail(YAGI_ERR_CONFIG, "block length %zu not supported (2n <= %d)", nn, 1 << 22);
        YG_TRY(require_device());
        h.assign(hh, hh + h_len);
        n = (int)nn;
        scale = div_scalar(one_of<C>(), 2.0f * (float)n);
        use_conv = h_len <= 2049;
        if (use_conv) {
            fir.st = st;
            YG_TRY(fir.init(hh, h_len));
            fir.kernel_choice = 4;
            return YAGI_OK;
        }
        YG_TRY(fft_plan_init(fwd, 2 * nn, YAGI_FFT_FORWARD));
        YG_TRY(fft_plan_init(bwd, 2 * nn, YAGI_FFT_BACKWARD));
        std::vector<cf32> tb(2 * nn, cf32{0.f, 0.f});
        for (size_t i = 0; i < h_len && i < 2 * nn; ++i) tb[i] = cx_of(h[i]);
        YG_TRY(tbuf.alloc(2 * nn * sizeof(cf32)));
        YG_TRY(hfreq.alloc(2 * nn * sizeof(cf32)));
        YG_TRY(upload(tbuf.p, tb.data(), 2 * nn * sizeof(cf32), st));
        YG_TRY(launch_fft_batch(fwd.d, tbuf.as<cf32>(), hfreq.as<cf32>(), 1, st));
        YG_TRY(w[0].alloc(nn * sizeof(cf32)));
        YG_TRY(w[1].alloc(nn * sizeof(cf32)));
        scale = div_scalar(one_of<C>(), 2.0f * (float)n);
        return reset();
    }
    int reset() {
        if (use_conv) return fir.w.reset(st);
        YG_HIP(hipMemsetAsync(w[cur].p, 0, (size_t)n * sizeof(cf32), st));
        return YAGI_OK;
    }
    int blocks_dev(const T *x, size_t nblocks, T *y) {
        if (nblocks == 0) return YAGI_OK;
        if (use_conv) {
            fir.st = st;
            fir.scale = mul_scalar(scale, 2.0f * (float)n);         // the caller's scale (ours is stored / 2n)
            return fir.block_dev(x, nblocks * (size_t)n, y);
        }
        const size_t n2 = 2 * (size_t)n;
        YG_TRY(tbuf.ensure(nblocks * n2 * sizeof(cf32)));
        YG_TRY(fbuf.ensure(nblocks * n2 * sizeof(cf32)));
        YG_TRY(launch_fftfilt_pad<T>(x, n, nblocks, tbuf.as<cf32>(), st));
        YG_TRY(launch_fft_batch(fwd.d, tbuf.as<cf32>(), fbuf.as<cf32>(), nblocks, st));
        YG_TRY(launch_fftfilt_mul(fbuf.as<cf32>(), hfreq.as<cf32>(), (int)n2, nblocks, st));
        YG_TRY(launch_fft_batch(bwd.d, fbuf.as<cf32>(), tbuf.as<cf32>(), nblocks, st));
        YG_TRY((launch_fftfilt_ola<T, C>(tbuf.as<cf32>(), w[cur].as<cf32>(), n, nblocks, scale, y,
                                         w[1 - cur].as<cf32>(), st)));
        cur = 1 - cur;
        return YAGI_OK;
    }
    int blocks_host(const T *x, size_t nblocks, T *y) {
        if (nblocks == 0) return YAGI_OK;
        const size_t bytes = nblocks * (size_t)n * sizeof(T);
        YG_TRY(ws.x.ensure(bytes));
        YG_TRY(ws.y.ensure(bytes));
        YG_TRY(upload(ws.x.p, x, bytes, st));
        YG_TRY(blocks_dev(ws.x.as<T>(), nblocks, ws.y.as<T>()));
        return download(y, ws.y.p, bytes, st);
    }
};

}  // namespace yagi

#define YAGI_FFTFILT_IMPL(K, KT, T, C)                                                              \
    struct yagi_hip_fftfilt_##K##_s : FftFiltObj<KT> {};                                            \
    extern "C" {                                                                                    \
    int yagi_hip_fftfilt_##K##_create(const C *h, size_t h_len, size_t n, yagi_hip_fftfilt_##K *q) try { \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (h_len && !h) return fail(YAGI_ERR_CONFIG, "null pointer argument");                     \
        auto o = std::make_unique<yagi_hip_fftfilt_##K##_s>();                                      \
        YG_TRY(o->init(h, h_len, n));                                                               \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_destroy(yagi_hip_fftfilt_##K q) try {                                \
        delete q;                                                                                   \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_clone(yagi_hip_fftfilt_##K q, yagi_hip_fftfilt_##K *out) try {       \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(out);                                                                             \
        *out = nullptr;                                                                             \
        auto o = std::make_unique<yagi_hip_fftfilt_##K##_s>();                                      \
        o->st = q->st;                                                                              \
        YG_TRY(o->init(q->h.data(), q->h.size(), (size_t)q->n));                                    \
        o->scale = q->scale;                                                                        \
        if (q->use_conv) {                                                                          \
            YG_TRY(q->fir.w.ensure_dev(q->st));                                                          \
            YG_TRY(o->fir.w.clone_from(q->fir.w, q->st));                                           \
        } else {                                                                                    \
            YG_HIP(hipMemcpyAsync(o->w[o->cur].p, q->w[q->cur].p, (size_t)q->n * sizeof(cf32),      \
                                  hipMemcpyDeviceToDevice, q->st));                                 \
        YG_HIP(hipStreamSynchronize(q->st));                                                    \
        }                                                                                           \
        *out = o.release();                                                                         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_set_stream(yagi_hip_fftfilt_##K q, yagi_stream_t s) try {            \
        CHECK_Q(q);                                                                                 \
        if (q->st == to_stream(s)) return YAGI_OK;                                                  \
        YG_HIP(hipStreamSynchronize(q->st));                                                        \
        q->st = to_stream(s);                                                                       \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_reset(yagi_hip_fftfilt_##K q) try {                                  \
        CHECK_Q(q);                                                                                 \
        return q->reset();                                                                          \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_set_scale(yagi_hip_fftfilt_##K q, C scale) try {                     \
        CHECK_Q(q);                                                                                 \
        q->scale = div_scalar(scale, 2.0f * (float)q->n);                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_get_scale(yagi_hip_fftfilt_##K q, C *scale) try {                    \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(scale);                                                                           \
        *scale = mul_scalar(q->scale, 2.0f * (float)q->n);                                          \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_get_length(yagi_hip_fftfilt_##K q, size_t *h_len) try {              \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(h_len);                                                                           \
        *h_len = q->h.size();                                                                       \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_execute(yagi_hip_fftfilt_##K q, const T *x, size_t nx, T *y,         \
                                       size_t ny) try {                                             \
        CHECK_Q(q);                                                                                 \
        if (nx != (size_t)q->n || ny != (size_t)q->n)                                               \
            return fail(YAGI_ERR_CONFIG, "input and output lengths must match filter block size");  \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        return q->blocks_host(x, 1, y);                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_execute_blocks(yagi_hip_fftfilt_##K q, const T *x, size_t nblocks,   \
                                              T *y) try {                                           \
        CHECK_Q(q);                                                                                 \
        if (nblocks == 0) return YAGI_OK;                                                           \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        return q->blocks_host(x, nblocks, y);                                                       \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_fftfilt_##K##_execute_blocks_dev(yagi_hip_fftfilt_##K q, const T *x,               \
                                                  size_t nblocks, T *y) try {                       \
        CHECK_Q(q);                                                                                 \
        if (nblocks == 0) return YAGI_OK;                                                           \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, nblocks * (size_t)q->n, y, nblocks * (size_t)q->n);                        \
        return q->blocks_dev(x, nblocks, y);                                                        \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    }

YAGI_FFTFILT_IMPL(rrrf, RRRF, float, float)
YAGI_FFTFILT_IMPL(crcf, CRCF, yagi_cf32, float)
YAGI_FFTFILT_IMPL(cccf, CCCF, yagi_cf32, yagi_cf32)

// ---- fused firfilt_crcf -> FFT stream ------------------------------------------------------------------
struct yagi_hip_firfft_crcf_s : FirFft {};

extern "C" {

int yagi_hip_firfft_crcf_create(const float *h, size_t h_len, size_t nfft, yagi_hip_firfft_crcf *q) try {
    CHECK_PTR(q);
    *q = nullptr;
    if (h_len && !h) return fail(YAGI_ERR_CONFIG, "null pointer argument");
    if (nfft == 0) return fail(YAGI_ERR_CONFIG, "fft length must be greater than zero");
    auto o = std::make_unique<yagi_hip_firfft_crcf_s>();
    YG_TRY(o->fir.init(h, h_len));
    if (o->fir.L > 2049) return fail(YAGI_ERR_CONFIG, "fused stream: filter too long (%zu taps, at most 2049)", h_len);
    o->nfft = nfft;
    if (nfft == 4096) YG_TRY(make_stream_twiddles(o->tw));
    else YG_TRY(fft_plan_init(o->plan, nfft, YAGI_FFT_FORWARD));      // any size the Fft object supports
    *q = o.release();
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_destroy(yagi_hip_firfft_crcf q) try { delete q; return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_set_pipeline(yagi_hip_firfft_crcf q, int on) try {
    CHECK_Q(q);
    YG_TRY(q->join());
    if (on) YG_TRY(q->fir.w.pipe.init());
    q->fir.w.pipe.on = on != 0;
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_join(yagi_hip_firfft_crcf q) try {
    CHECK_Q(q);
    return q->join();
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_set_stream(yagi_hip_firfft_crcf q, yagi_stream_t s) try {
    CHECK_Q(q);
    if (q->fir.st == to_stream(s)) return YAGI_OK;
    YG_TRY(q->join());
    YG_HIP(hipStreamSynchronize(q->fir.st));
    q->fir.st = to_stream(s);
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_set_scale(yagi_hip_firfft_crcf q, float scale) try { CHECK_Q(q); q->fir.scale = scale; return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_reset(yagi_hip_firfft_crcf q) try {
    CHECK_Q(q);
    YG_TRY(q->join());
    return q->fir.w.reset(q->fir.st);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_set_variant(yagi_hip_firfft_crcf q, int variant) try {
    CHECK_Q(q);
    if (variant < 0 || variant > 4) return fail(YAGI_ERR_CONFIG, "unknown variant %d", variant);
    if (q->nfft != 4096 && variant != 0 && variant != 3)
        return fail(YAGI_ERR_CONFIG, "nfft = %zu runs as overlap-save FIR + batched FFT (variant 3) only", q->nfft);
    if (variant == 3 && q->fir.L > 2049) return fail(YAGI_ERR_CONFIG, "fast convolution needs <= 2049 taps");
    if (variant == 4 && q->fir.L > 257) return fail(YAGI_ERR_CONFIG, "frequency-domain variant needs <= 257 taps");
    if (variant == 2 && !q->fir.Lm) return fail(YAGI_ERR_CONFIG, "MFMA variant needs <= 256 taps");
    if (variant == 1 && q->fir.Lp > kSlideMaxTaps) return fail(YAGI_ERR_CONFIG, "sliding variant needs <= %d taps", kSlideMaxTaps);
    YG_TRY(q->join());
    q->variant = variant;
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_execute_dev(yagi_hip_firfft_crcf q, const yagi_cf32 *x, size_t nframes, yagi_cf32 *spectra) try {
    CHECK_Q(q);
    if (nframes == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(spectra);
    CHECK_NOALIAS(x, nframes * q->nfft, spectra, nframes * q->nfft);
    auto &f = q->fir;
    const bool use_freq = q->nfft == 4096 && (q->variant == 4 || (q->variant == 0 && f.L <= 257));
    const bool piped = use_freq && f.w.can_pipe(nframes * q->nfft) && f.conv_ready && f.hfreq_s_valid &&
                       f.hfreq_s_scale == f.scale;
    if (!piped) YG_TRY(f.w.ensure_dev(f.st));   // joins the lanes: everything below runs on the caller's stream
    // auto (0): fast convolution once the filter is long enough for it to win (measured crossover ~100 taps:
    // the direct kernels cost ~0.65 us per tap and 2^24 samples, the convolution kernel a flat 88 us)
    if (q->nfft != 4096) {     // other frame lengths: overlap-save FIR into a scratch stream, then the batched transform
        YG_TRY(f.prepare_conv());
        const size_t n = nframes * q->nfft;
        YG_TRY(q->scratch.ensure(n * sizeof(cf32)));
        cf32 *ys = q->scratch.as<cf32>();
        YG_TRY(launch_fir_crcf_fftconv(f.w.dev(), x, 0, n, f.hfreq.as<cf32>(), f.scale, f.L, f.twf.as<cf32>(),
                                       f.twb.as<cf32>(), ys, n, f.st));
        YG_TRY(launch_fft_batch(q->plan.d, ys, spectra, nframes, f.st));
        return f.w.advance(x, n, f.st);
    }
    if (piped) {
        // the frequency-domain kernel on the next lane (see StreamPipe); tables and scaled FFT{h} are in place
        const size_t n = nframes * q->nfft;
        hipStream_t lane;
        const cf32 *win;
        YG_TRY(f.w.begin_piped(f.st, &lane, &win));
        YG_TRY(launch_firfft_crcf_4096_freq(win, x, f.hfreq_s.as<cf32>(), f.gfft_s.as<cf32>(), f.L, q->tw.as<cf32>(), spectra,
                                            nullptr, nframes, lane));
        return f.w.finish_piped(x, n);
    }
    if (use_freq) {
        // frequency-domain form (one kernel, 16 B/sample): FFT{h}.FFT{x_f} + FFT{frame-boundary correction}
        YG_TRY(f.prepare_conv());
        if (!f.hfreq_s_valid || f.hfreq_s_scale != f.scale) {
            YG_TRY(f.hfreq_s.ensure(4096 * sizeof(cf32)));
            YG_TRY(launch_scale_cf32(f.hfreq.as<cf32>(), f.scale, f.hfreq_s.as<cf32>(), 4096, f.st));
            YG_TRY(f.gfft_s.ensure(512 * sizeof(cf32)));
            YG_TRY(launch_scale_cf32(f.gfft.as<cf32>(), f.scale, f.gfft_s.as<cf32>(), 512, f.st));
            f.hfreq_s_valid = true;
            f.hfreq_s_scale = f.scale;
        }
        YG_TRY(launch_firfft_crcf_4096_freq(f.w.dev(), x, f.hfreq_s.as<cf32>(), f.gfft_s.as<cf32>(), f.L,
                                            q->tw.as<cf32>(), spectra, f.w.next(), nframes, f.st));
        f.w.flip();
        return YAGI_OK;
    }
    const bool use_conv = q->variant == 3 || (q->variant == 0 && f.L <= 2049);
    if (use_conv) {
        // fast-convolution form (two kernels): overlap-save FIR into a scratch stream, then the batched
        // 4096-point FFT over its frames.  32 B/sample of HBM traffic instead of 16.  (Running the two
        // kernels as a chunked two-stream pipeline was measured slower: profiles/r01_notes.md.)
        YG_TRY(f.prepare_conv());
        const size_t n = nframes * q->nfft;
        YG_TRY(q->scratch.ensure(n * sizeof(cf32)));
        FftPlanDev d;
        d.n = 4096;
        d.dir = YAGI_FFT_FORWARD;
        d.tw = q->tw.as<cf32>();
        cf32 *ys = q->scratch.as<cf32>();
        YG_TRY(launch_fir_crcf_fftconv(f.w.dev(), x, 0, n, f.hfreq.as<cf32>(), f.scale, f.L, f.twf.as<cf32>(),
                                       f.twb.as<cf32>(), ys, n, f.st));
        YG_TRY(launch_fft_batch(d, ys, spectra, nframes, f.st));
        return f.w.advance(x, n, f.st);
    }
    YG_TRY(launch_firfft_crcf_4096(f.w.dev(), x, f.taps_pad.as<float>(), f.apack.as<float>(), f.L, f.Lp, f.Lm, f.scale,
                                   q->tw.as<cf32>(), spectra, nframes, q->variant, f.st));
    return f.w.advance(x, nframes * q->nfft, f.st);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firfft_crcf_execute(yagi_hip_firfft_crcf q, const yagi_cf32 *x, size_t nframes, yagi_cf32 *spectra) try {
    CHECK_Q(q);
    if (nframes == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(spectra);
    const size_t bytes = nframes * q->nfft * sizeof(cf32);
    YG_TRY(q->join());
    YG_TRY(q->xin.ensure(bytes));
    YG_TRY(q->yout.ensure(bytes));
    YG_TRY(upload(q->xin.p, x, bytes, q->fir.st));
    YG_TRY(yagi_hip_firfft_crcf_execute_dev(q, q->xin.as<cf32>(), nframes, q->yout.as<cf32>()));
    YG_TRY(q->join());
    return download(spectra, q->yout.p, bytes, q->fir.st);
} catch (...) { return ::yagi::api_exception(); }

}  // extern "C"

// ---- channelizers ------------------------------------------------------------------------------------
struct yagi_hip_firpfbch_crcf_s : PfbCh {};
struct yagi_hip_firpfbch2_crcf_s : PfbCh2 {};

extern "C" {

int yagi_hip_firpfbch_crcf_create(size_t M, size_t p, const float *h, yagi_hip_firpfbch_crcf *q) try {
    CHECK_PTR(q);
    *q = nullptr;
    if (M == 0) return fail(YAGI_ERR_CONFIG, "number of channels must be greater than zero");
    if (p == 0) return fail(YAGI_ERR_CONFIG, "invalid filter size (must be greater than 0)");
    CHECK_PTR(h);
    if (M > 8192 || p > 4096) return fail(YAGI_ERR_CONFIG, "channelizer too large");
    YG_TRY(require_device());
    auto o = std::make_unique<yagi_hip_firpfbch_crcf_s>();
    o->M = (int)M;
    o->p = (int)p;
    YG_TRY(o->h.alloc(M * p * sizeof(float)));
    YG_TRY(upload(o->h.p, h, M * p * sizeof(float), nullptr));
    YG_TRY(make_twiddles((int)M, YAGI_FFT_FORWARD, o->tw));
    const size_t hl = (p - 1) * M;
    YG_TRY(o->hist.init((int)(hl ? hl : 1), nullptr));
    YG_TRY(o->syn_hist.init((int)(hl ? hl : 1), nullptr));
    *q = o.release();
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch_crcf_create_kaiser(size_t M, size_t m, float as_, yagi_hip_firpfbch_crcf *q) try {
    CHECK_PTR(q);
    *q = nullptr;
    if (M == 0) return fail(YAGI_ERR_CONFIG, "number of channels must be greater than zero");
    if (m == 0) return fail(YAGI_ERR_CONFIG, "invalid filter size (must be greater than 0)");
    std::vector<float> hf;
    YG_TRY(design_kaiser(2 * M * m + 1, 0.5f / (float)M, std::fabs(as_), 0.0f, hf));
    return yagi_hip_firpfbch_crcf_create(M, 2 * m, hf.data(), q);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch_crcf_destroy(yagi_hip_firpfbch_crcf q) try { delete q; return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch_crcf_set_stream(yagi_hip_firpfbch_crcf q, yagi_stream_t s) try {
    CHECK_Q(q);
    if (q->st == to_stream(s)) return YAGI_OK;      // unchanged: no host synchronisation
    YG_HIP(hipStreamSynchronize(q->st));
    q->st = to_stream(s);
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch_crcf_reset(yagi_hip_firpfbch_crcf q) try {
    CHECK_Q(q);
    YG_TRY(q->hist.reset(q->st));
    return q->syn_hist.reset(q->st);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch_crcf_analyzer_execute_dev(yagi_hip_firpfbch_crcf q, const yagi_cf32 *x, size_t nframes, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (nframes == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    CHECK_NOALIAS(x, nframes * (size_t)q->M, y, nframes * (size_t)q->M);
    // a p == 1 channelizer has no history; the 1-sample placeholder window is never read
    bool written = false;
    cf32 *next = (q->p > 1 && q->hist.len == (int)((q->p - 1) * q->M)) ? q->hist.next() : nullptr;
    YG_TRY(launch_firpfbch(q->hist.dev(), x, q->h.as<float>(), q->M, q->p, q->tw.as<cf32>(), y, nframes, q->st, next, &written));
    if (written) { q->hist.flip(); return YAGI_OK; }         // the kernel's last workgroup wrote the next history
    if (q->p > 1) return q->hist.advance(x, nframes * (size_t)q->M, q->st);
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch_crcf_analyzer_execute(yagi_hip_firpfbch_crcf q, const yagi_cf32 *x, size_t nframes, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (nframes == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    const size_t bytes = nframes * (size_t)q->M * sizeof(cf32);
    YG_TRY(q->ws.x.ensure(bytes));
    YG_TRY(q->ws.y.ensure(bytes));
    YG_TRY(upload(q->ws.x.p, x, bytes, q->st));
    YG_TRY(yagi_hip_firpfbch_crcf_analyzer_execute_dev(q, q->ws.x.as<cf32>(), nframes, q->ws.y.as<cf32>()));
    return download(y, q->ws.y.p, bytes, q->st);
} catch (...) { return ::yagi::api_exception(); }

int yagi_hip_firpfbch_crcf_synthesizer_execute_dev(yagi_hip_firpfbch_crcf q, const yagi_cf32 *x, size_t nframes, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (nframes == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    CHECK_NOALIAS(x, nframes * (size_t)q->M, y, nframes * (size_t)q->M);
    YG_TRY(launch_firpfbch_syn(q->syn_hist.dev(), x, q->h.as<float>(), q->M, q->p, q->tw.as<cf32>(), y, nframes, q->st));
    if (q->p > 1) return q->syn_hist.advance(x, nframes * (size_t)q->M, q->st);
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch_crcf_synthesizer_execute(yagi_hip_firpfbch_crcf q, const yagi_cf32 *x, size_t nframes, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (nframes == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    const size_t bytes = nframes * (size_t)q->M * sizeof(cf32);
    YG_TRY(q->ws.x.ensure(bytes));
    YG_TRY(q->ws.y.ensure(bytes));
    YG_TRY(upload(q->ws.x.p, x, bytes, q->st));
    YG_TRY(yagi_hip_firpfbch_crcf_synthesizer_execute_dev(q, q->ws.x.as<cf32>(), nframes, q->ws.y.as<cf32>()));
    return download(y, q->ws.y.p, bytes, q->st);
} catch (...) { return ::yagi::api_exception(); }

int yagi_hip_firpfbch2_crcf_create(size_t M, size_t m, const float *h, yagi_hip_firpfbch2_crcf *q) try {
    CHECK_PTR(q);
    *q = nullptr;
    if (M < 2 || (M & 1)) return fail(YAGI_ERR_CONFIG, "number of channels must be greater than 2 and even");
    if (m < 1) return fail(YAGI_ERR_CONFIG, "filter semi-length must be at least 1");
    CHECK_PTR(h);
    if (M > 8192 || m > 2048) return fail(YAGI_ERR_CONFIG, "channelizer too large");
    YG_TRY(require_device());
    auto o = std::make_unique<yagi_hip_firpfbch2_crcf_s>();
    o->M = (int)M;
    o->m = (int)m;
    const size_t hl = 2 * M * m;
    YG_TRY(o->h.alloc(hl * sizeof(float)));
    YG_TRY(upload(o->h.p, h, hl * sizeof(float), nullptr));
    YG_TRY(make_twiddles((int)M, YAGI_FFT_FORWARD, o->tw));
    YG_TRY(o->hist.init((int)((2 * m - 1) * M + M / 2), nullptr));
    YG_TRY(o->syn_hist.init((int)((4 * m - 1) * M), nullptr));
    *q = o.release();
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_create_kaiser(size_t M, size_t m, float as_, yagi_hip_firpfbch2_crcf *q) try {
    CHECK_PTR(q);
    *q = nullptr;
    if (M < 2 || (M & 1)) return fail(YAGI_ERR_CONFIG, "number of channels must be greater than 2 and even");
    if (m < 1) return fail(YAGI_ERR_CONFIG, "filter semi-length must be at least 1");
    std::vector<float> hf;
    YG_TRY(design_kaiser(2 * M * m + 1, 1.0f / (float)M, std::fabs(as_), 0.0f, hf));
    float hsum = 0.0f;                          // normalise to unit channel gain: sum(h) = M
    for (float v : hf) hsum += v;
    for (float &v : hf) v = v * (float)M / hsum;
    return yagi_hip_firpfbch2_crcf_create(M, m, hf.data(), q);
} catch (...) { return ::yagi::api_exception(); }
// prototype of the matching synthesizer: kaiser(2Mm+1, 0.5/M, as), scaled to sum M (analyzer: cutoff 1/M)
int yagi_hip_firpfbch2_crcf_create_kaiser_synthesizer(size_t M, size_t m, float as_, yagi_hip_firpfbch2_crcf *q) try {
    CHECK_PTR(q);
    *q = nullptr;
    if (M < 2 || (M & 1)) return fail(YAGI_ERR_CONFIG, "number of channels must be greater than 2 and even");
    if (m < 1) return fail(YAGI_ERR_CONFIG, "filter semi-length must be at least 1");
    std::vector<float> hf;
    YG_TRY(design_kaiser(2 * M * m + 1, 0.5f / (float)M, std::fabs(as_), 0.0f, hf));
    float hsum = 0.0f;
    for (float v : hf) hsum += v;
    for (float &v : hf) v = v * (float)M / hsum;
    return yagi_hip_firpfbch2_crcf_create(M, m, hf.data(), q);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_destroy(yagi_hip_firpfbch2_crcf q) try { delete q; return YAGI_OK; } catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_set_stream(yagi_hip_firpfbch2_crcf q, yagi_stream_t s) try {
    CHECK_Q(q);
    if (q->st == to_stream(s)) return YAGI_OK;      // unchanged: no host synchronisation
    YG_HIP(hipStreamSynchronize(q->st));
    q->st = to_stream(s);
    return YAGI_OK;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_reset(yagi_hip_firpfbch2_crcf q) try {
    CHECK_Q(q);
    q->step = 0;
    q->syn_step = 0;
    YG_TRY(q->hist.reset(q->st));
    return q->syn_hist.reset(q->st);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_synthesizer_execute_dev(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x, size_t nsteps, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (nsteps == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    CHECK_NOALIAS(x, nsteps * (size_t)q->M, y, nsteps * (size_t)(q->M / 2));
    YG_TRY(launch_firpfbch2_syn(q->syn_hist.dev(), q->syn_hist.len, x, q->h.as<float>(), q->M, q->m, q->tw.as<cf32>(),
                                q->syn_step, y, nsteps, q->st));
    q->syn_step += nsteps;
    return q->syn_hist.advance(x, nsteps * (size_t)q->M, q->st);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_synthesizer_execute(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x, size_t nsteps, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (nsteps == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    const size_t in_bytes = nsteps * (size_t)q->M * sizeof(cf32), out_bytes = in_bytes / 2;
    YG_TRY(q->ws.x.ensure(in_bytes));
    YG_TRY(q->ws.y.ensure(out_bytes));
    YG_TRY(upload(q->ws.x.p, x, in_bytes, q->st));
    YG_TRY(yagi_hip_firpfbch2_crcf_synthesizer_execute_dev(q, q->ws.x.as<cf32>(), nsteps, q->ws.y.as<cf32>()));
    return download(y, q->ws.y.p, out_bytes, q->st);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_analyzer_execute_shard_dev(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x, size_t nsteps,
                                                       int rank, int nranks, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (nsteps == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    if (nranks > 0) CHECK_NOALIAS(x, nsteps * (size_t)(q->M / 2), y, nsteps * (size_t)(q->M / nranks));
    bool written = false;
    YG_TRY(launch_firpfbch2(q->hist.dev(), q->hist.len, x, q->h.as<float>(), q->M, q->m, q->tw.as<cf32>(),
                            q->step, rank, nranks, y, nsteps, q->st, q->hist.next(), &written));
    q->step += nsteps;
    if (written) { q->hist.flip(); return YAGI_OK; }         // the kernel's last workgroup wrote the next history
    return q->hist.advance(x, nsteps * (size_t)(q->M / 2), q->st);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_analyzer_execute_dev(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x, size_t nsteps, yagi_cf32 *y) try {
    return yagi_hip_firpfbch2_crcf_analyzer_execute_shard_dev(q, x, nsteps, 0, 1, y);
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_analyzer_execute(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x, size_t nsteps, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (nsteps == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    const size_t bin = nsteps * (size_t)(q->M / 2) * sizeof(cf32), bout = nsteps * (size_t)q->M * sizeof(cf32);
    YG_TRY(q->ws.x.ensure(bin));
    YG_TRY(q->ws.y.ensure(bout));
    YG_TRY(upload(q->ws.x.p, x, bin, q->st));
    YG_TRY(yagi_hip_firpfbch2_crcf_analyzer_execute_dev(q, q->ws.x.as<cf32>(), nsteps, q->ws.y.as<cf32>()));
    return download(y, q->ws.y.p, bout, q->st);
} catch (...) { return ::yagi::api_exception(); }
// sub-bands sharded over the ranks of `comm`: shard kernel (this stream) -> RCCL all-gather -> assemble (the
// communicator's stream), chunk by chunk so chunk k's exchange runs beside chunk k+1's kernel
int yagi_hip_firpfbch2_crcf_analyzer_execute_sharded_dev(yagi_hip_firpfbch2_crcf q, const yagi_cf32 *x, size_t nsteps,
                                                         yagi_hip_comm comm, int nchunks, yagi_cf32 *y) try {
    CHECK_Q(q);
    if (!comm) return fail(YAGI_ERR_CONFIG, "null communicator");
    if (nsteps == 0) return YAGI_OK;
    CHECK_PTR(x);
    CHECK_PTR(y);
    CHECK_NOALIAS(x, nsteps * (size_t)(q->M / 2), y, nsteps * (size_t)q->M);
    const int R = comm->nranks, rank = comm->rank;
    if (R == 1 && nchunks >= 0) return yagi_hip_firpfbch2_crcf_analyzer_execute_dev(q, x, nsteps, y);
    if (q->M % R) return fail(YAGI_ERR_CONFIG, "firpfbch2: %d channels do not shard over %d ranks", q->M, R);
    const size_t M = (size_t)q->M, Mr = M / (size_t)R, M2 = M / 2;
    // chunks: even step counts (the column kernels start on an even step), at least 2048 steps each
    size_t nc = nchunks ? (size_t)(nchunks < 0 ? -nchunks : nchunks) : 8;
    while (nc > 1 && nsteps / nc < 2048) --nc;
    size_t per = (nsteps + nc - 1) / nc;
    per += per & 1;
    YG_TRY(q->shard.ensure(nsteps * Mr * sizeof(cf32)));
    YG_TRY(q->gathered.ensure(nsteps * M * sizeof(cf32)));
    while (comm->ev.size() < nc) {
        hipEvent_t e;
        YG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        comm->ev.push_back(e);
    }
    // the exchange stream must not run ahead of work already queued on the object's stream that still reads y
    auto chunks = [&]() -> int {
        size_t done = 0, k = 0;
        while (done < nsteps) {
            const size_t ns = std::min(per, nsteps - done);
            cf32 *sh = q->shard.as<cf32>() + done * Mr;
            cf32 *ga = q->gathered.as<cf32>() + done * M;
            YG_TRY(yagi_hip_firpfbch2_crcf_analyzer_execute_shard_dev(q, x + done * M2, ns, rank, R, sh));
            YG_HIP(hipEventRecord(comm->ev[k], q->st));
            YG_HIP(hipStreamWaitEvent(comm->st, comm->ev[k], 0));
            YG_TRY(comm_all_gather(comm, sh, ga, ns * Mr * sizeof(cf32), comm->st));
            YG_TRY(launch_firpfbch2_assemble(ga, ns, (int)M, R, y + done * M, comm->st));
            done += ns;
            ++k;
        }
        return YAGI_OK;
    };
    const int rc = chunks();
    // success or not: the object's stream waits for whatever the exchange stream was handed, so a later call cannot
    // rewrite (or ensure() free) q->shard / q->gathered while a gather or an assemble still reads them
    if (hipEventRecord(comm->done, comm->st) != hipSuccess || hipStreamWaitEvent(q->st, comm->done, 0) != hipSuccess) {
        (void)hipStreamSynchronize(comm->st);
        if (rc == YAGI_OK) return fail(YAGI_ERR_DEVICE, "joining the exchange stream failed");
    }
    return rc;
} catch (...) { return ::yagi::api_exception(); }
int yagi_hip_firpfbch2_crcf_assemble_dev(const yagi_cf32 *gathered, size_t nsteps, size_t M, int nranks,
                                         yagi_cf32 *y, yagi_stream_t s) try {
    if (nsteps == 0) return YAGI_OK;
    CHECK_PTR(gathered);
    CHECK_PTR(y);
    return launch_firpfbch2_assemble(gathered, nsteps, (int)M, nranks, y, to_stream(s));
} catch (...) { return ::yagi::api_exception(); }

}  // extern "C"


// ---- Resamp2 / MsResamp2 (src/filter/resampler/resamp2.rs, msresamp2.rs) ---------------
namespace yagi {

// Resamp2Coeff::for_halfband (resamp2.rs:9-23), f32 arithmetic as the reference spells it
static float for_halfband(float hf, float t, float f0, float *) {
    const float pi = 3.14159265358979323846f;
    return 2.0f * hf * std::cos(2.0f * pi * t * f0);
}
static cf32 for_halfband(float hf, float t, float f0, cf32 *) {
    const float pi = 3.14159265358979323846f;
    const float g = 2.0f * hf, a = 2.0f * pi * t * f0;
    return cf32{g * std::cos(a), g * std::sin(a)};
}

template <class K>
struct Resamp2Obj {
    using T = typename K::T;
    using C = typename K::C;
    hipStream_t st = nullptr;
    int m = 0;
    std::vector<C> h1;             // h1[i] = h[h_len - 2i - 2] (resamp2.rs:66-70)
    DevBuf h1d;
    DevBuf state[2];               // [w0 (2m, oldest first)][w1 (2m)], ping-pong
    int cur = 0;
    int toggle = 0;
    C scale;
    Workspace ws;

    // new() from the designed half-band prototype hf[4m+1] (:44-88)
    int init(const float *hf, size_t m_, float f0) {
        if (m_ < 2) return fail(YAGI_ERR_CONFIG, "filter semi-length must be at least 2");
        if (m_ > (size_t)kR2MaxSemiLen) return fail(YAGI_ERR_CONFIG, "filter semi-length must not exceed %d", kR2MaxSemiLen);
        if (f0 < -0.5f || f0 > 0.5f) return fail(YAGI_ERR_CONFIG, "f0 (%g) must be in [-0.5,0.5]", (double)f0);
        m = (int)m_;
        const size_t h_len = 4 * m_ + 1;
        std::vector<C> h(h_len);
        for (size_t i = 0; i < h_len; ++i) {
            const float t = (float)i - (float)(h_len - 1) / 2.0f;
            h[i] = for_halfband(hf[i], t, f0, (C *)nullptr);
        }
        h1.resize(2 * m_);
        for (size_t i = 0; i < 2 * m_; ++i) h1[i] = h[h_len - 2 * i - 2];
        scale = to_c(1.0f, (C *)nullptr);
        YG_TRY(h1d.alloc(h1.size() * sizeof(C)));
        YG_TRY(upload(h1d.p, h1.data(), h1.size() * sizeof(C), st));
        YG_TRY(state[0].alloc(4 * m_ * sizeof(T)));
        YG_TRY(state[1].alloc(4 * m_ * sizeof(T)));
        return reset();
    }
    int reset() {                                                   // :90-94
        toggle = 0;
        YG_HIP(hipMemsetAsync(state[cur].p, 0, 4 * (size_t)m * sizeof(T), st));
        return YAGI_OK;
    }
    static size_t out_count(int mode, size_t nx) {
        return mode == kR2Filter || mode == kR2Interp ? 2 * nx : mode == kR2Decim ? nx / 2 : nx;
    }
    // the reference's slices carry their lengths (copy_from_slice / indexing panics on a mismatch); the C ABI checks
    static int check_lengths(int mode, size_t nx, size_t ny) {
        if (mode < kR2Filter || mode > kR2Interp) return fail(YAGI_ERR_CONFIG, "resamp2: unknown form %d", mode);
        if (mode != kR2Filter && mode != kR2Interp && (nx & 1))
            return fail(YAGI_ERR_CONFIG, "resamp2: this form consumes pairs of samples (got %zu)", nx);
        if (ny != out_count(mode, nx))
            return fail(YAGI_ERR_CONFIG, "resamp2: form %d turns %zu samples into %zu, the output holds %zu", mode, nx,
                        out_count(mode, nx), ny);
        return YAGI_OK;
    }
    int block_dev(int mode, const T *x, size_t nx, T *y) {
        if (mode < kR2Filter || mode > kR2Interp) return fail(YAGI_ERR_CONFIG, "resamp2: unknown form %d", mode);
        if (mode != kR2Filter && mode != kR2Interp && (nx & 1))
            return fail(YAGI_ERR_CONFIG, "resamp2: this form consumes pairs of samples (got %zu)", nx);
        if (nx == 0) return YAGI_OK;
        if (mode == kR2Decim && nx >= ((size_t)1 << 19) && m <= 64) {
            // a long decimator block: the one-stage case of the MsResamp2 chain kernels (the same sums in the same order;
            // the middle of the block four outputs per lane through msresamp2_decim_fast_kernel)
            const int mk = m;
            const C sc = scale;
            const C *hp = h1d.template as<C>();
            const T *sp = state[cur].template as<T>();
            T *sn = state[1 - cur].template as<T>();
            YG_TRY((launch_msresamp2_decim<T, C>(1, &mk, &sc, &hp, &sp, &sn, x, y, nx / 2, st)));
        } else {
            YG_TRY((launch_resamp2<T, C>(mode, state[cur].template as<T>(), x, nx, h1d.template as<C>(), m, scale, toggle, y,
                                         state[1 - cur].template as<T>(), st)));
        }
        cur = 1 - cur;
        if (mode == kR2Filter) toggle = (toggle + (int)(nx & 1)) & 1;
        return YAGI_OK;
    }
    int block_host(int mode, const T *x, size_t nx, T *y) {
        if (nx == 0) return YAGI_OK;
        const size_t ny = out_count(mode, nx);
        YG_TRY(ws.x.ensure(nx * sizeof(T)));
        YG_TRY(ws.y.ensure((ny ? ny : 1) * sizeof(T)));
        YG_TRY(upload(ws.x.p, x, nx * sizeof(T), st));
        YG_TRY(block_dev(mode, ws.x.template as<T>(), nx, ws.y.template as<T>()));
        return download(y, ws.y.p, ny * sizeof(T), st);
    }
    int clone_into(Resamp2Obj &o) const {
        o.st = st;
        o.m = m;
        o.h1 = h1;
        o.toggle = toggle;
        o.scale = scale;
        o.cur = 0;
        YG_TRY(o.h1d.alloc(h1.size() * sizeof(C)));
        YG_TRY(upload(o.h1d.p, h1.data(), h1.size() * sizeof(C), st));
        YG_TRY(o.state[0].alloc(4 * (size_t)m * sizeof(T)));
        YG_TRY(o.state[1].alloc(4 * (size_t)m * sizeof(T)));
        YG_HIP(hipMemcpyAsync(o.state[0].p, state[cur].p, 4 * (size_t)m * sizeof(T), hipMemcpyDeviceToDevice, st));
        YG_HIP(hipStreamSynchronize(st));
        return YAGI_OK;
    }
};

// estimate_req_filter_len (design/mod.rs:138-152 with Kaiser's formula :228-238) and MsResamp2::new's stage plan
// (msresamp2.rs:66-88): semi-length of every half-band stage
static int msresamp2_stage_lengths(size_t num_stages, float fc, float as_, std::vector<size_t> &m_stage) {
    if (num_stages > 16) return fail(YAGI_ERR_CONFIG, "number of stages should not exceed 16");
    if (fc <= 0.0f || fc >= 0.5f) return fail(YAGI_ERR_CONFIG, "cut-off frequency must be in (0,0.5)");
    m_stage.assign(num_stages, 0);
    const float a = as_ + 5.0f;
    for (size_t i = 0; i < num_stages; ++i) {
        fc = (i == 1) ? (0.5f - fc) / 2.0f : 0.5f * fc;
        const float ft = 2.0f * (0.25f - fc);
        if (ft <= 0.0f || ft > 0.5f) return fail(YAGI_ERR_CONFIG, "cutoff frequency (%g) out of range (0, 0.5)", (double)ft);
        if (a <= 0.0f) return fail(YAGI_ERR_CONFIG, "stopband attenuation must be greater than zero");
        const float hl = (a - 7.95f) / (14.26f * ft);
        const size_t h_len = hl <= 0.0f ? 0 : (size_t)hl;
        const size_t mm = (size_t)std::ceil(((float)h_len - 1.0f) / 4.0f);
        m_stage[i] = mm < 3 ? 3 : mm;
    }
    return YAGI_OK;
}

template <class K>
struct MsResamp2Obj {
    using T = typename K::T;
    using C = typename K::C;
    hipStream_t st = nullptr;
    bool interp = false;
    size_t num_stages = 0, rate = 1;
    std::vector<size_t> m_stage;
    std::vector<std::unique_ptr<Resamp2Obj<K>>> stage;
    DevBuf buf[2];
    Workspace ws;

    // n execute() calls: interpolator n -> n * rate, decimator n * rate -> n (msresamp2.rs:137-152, :181 copy_from_slice)
    int check_lengths(size_t nx, size_t ny, size_t *n) const {
        const size_t big = interp ? ny : nx, small = interp ? nx : ny;
        if (big != small * rate)
            return fail(YAGI_ERR_CONFIG, "msresamp2: %s by %zu needs %zu %s samples for %zu %s samples (got %zu)",
                        interp ? "interpolation" : "decimation", rate, small * rate, interp ? "output" : "input", small,
                        interp ? "input" : "output", big);
        *n = small;
        return YAGI_OK;
    }

    int init(bool interp_, size_t ns, const size_t *ms, const float *hf_all) {
        if (ns > 16) return fail(YAGI_ERR_CONFIG, "number of stages should not exceed 16");
        interp = interp_;
        num_stages = ns;
        rate = (size_t)1 << ns;
        m_stage.assign(ms, ms + ns);
        for (size_t i = 0; i < ns; ++i) {
            auto s = std::make_unique<Resamp2Obj<K>>();
            s->st = st;
            YG_TRY(s->init(hf_all, ms[i], 0.0f));                    // f0_stage = 0 (msresamp2.rs:44-46)
            hf_all += 4 * ms[i] + 1;
            stage.push_back(std::move(s));
        }
        // the decimator's zeta = 1/rate multiplies the last stage's output (:197); folded into that stage's scale
        // (its scale is 1, so (y0 + y1) * 1 * zeta == (y0 + y1) * zeta bit for bit)
        if (!interp && ns) stage[0]->scale = to_c(1.0f / (float)rate, (C *)nullptr);
        return YAGI_OK;
    }
    // n execute() calls: interp n -> n*rate, decim n*rate -> n (device pointers)
    int block_dev(const T *x, size_t n, T *y) {
        if (n == 0) return YAGI_OK;
        if (num_stages == 0) {
            YG_HIP(hipMemcpyAsync(y, x, n * sizeof(T), hipMemcpyDeviceToDevice, st));
            return YAGI_OK;
        }
        const size_t big = n * rate;
        if (num_stages > 1) {                                        // ping-pong intermediates: at most big/2 samples
            YG_TRY(buf[0].ensure(big / 2 * sizeof(T)));
            YG_TRY(buf[1].ensure(big / 2 * sizeof(T)));
        }
        const T *src = x;
        int pp = 0;
        if (interp && num_stages <= 4 && fused_interp_fits()) {      // the whole chain in one launch (LDS-resident levels)
            const int ns = (int)num_stages;
            int mk[4];
            C sc[4];
            const C *h1[4];
            const T *sta[4];
            T *stn[4];
            for (int k = 0; k < ns; ++k) {                           // processing order: stage k
                Resamp2Obj<K> &o = *stage[(size_t)k];
                mk[k] = o.m;
                sc[k] = o.scale;
                h1[k] = o.h1d.template as<C>();
                sta[k] = o.state[o.cur].template as<T>();
                stn[k] = o.state[1 - o.cur].template as<T>();
            }
            YG_TRY((launch_msresamp2_interp<T, C>(ns, mk, sc, h1, sta, stn, x, y, n, st)));
            for (size_t g = 0; g < num_stages; ++g) stage[g]->cur = 1 - stage[g]->cur;
        } else if (interp) {                                         // stage s doubles n 2^s samples (:154-175)
            size_t cnt = n;
            for (size_t s = 0; s < num_stages; ++s) {
                const bool last = s + 1 == num_stages;
                T *dst = last ? y : buf[pp].template as<T>();
                YG_TRY(stage[s]->block_dev(kR2Interp, src, cnt, dst));
                src = dst;
                cnt *= 2;
                pp ^= 1;
            }
        } else if (num_stages <= 4 && fused_decim_fits()) {          // the whole chain in one launch (LDS-resident intermediates)
            const int ns = (int)num_stages;
            int mk[4];
            C sc[4];
            const C *h1[4];
            const T *sta[4];
            T *stn[4];
            for (int k = 0; k < ns; ++k) {                           // processing order: stage g = S-1-k
                Resamp2Obj<K> &o = *stage[num_stages - 1 - (size_t)k];
                mk[k] = o.m;
                sc[k] = o.scale;
                h1[k] = o.h1d.template as<C>();
                sta[k] = o.state[o.cur].template as<T>();
                stn[k] = o.state[1 - o.cur].template as<T>();
            }
            YG_TRY((launch_msresamp2_decim<T, C>(ns, mk, sc, h1, sta, stn, x, y, n, st)));
            for (size_t g = 0; g < num_stages; ++g) stage[g]->cur = 1 - stage[g]->cur;
        } else {                                                     // stages g = S-1 .. 0, each halves (:177-197)
            size_t cnt = big;
            for (size_t s = 0; s < num_stages; ++s) {
                const size_t g = num_stages - 1 - s;
                const bool last = g == 0;
                T *dst = last ? y : buf[pp].template as<T>();
                YG_TRY(stage[g]->block_dev(kR2Decim, src, cnt, dst));
                src = dst;
                cnt /= 2;
                pp ^= 1;
            }
        }
        return YAGI_OK;
    }
    bool fused_interp_fits() const {
        int mk[4];
        for (size_t k = 0; k < num_stages && k < 4; ++k) mk[k] = stage[k]->m;
        return msresamp2_interp_lds((int)num_stages, mk, sizeof(T)) <= 64 * 1024;
    }
    bool fused_decim_fits() const {
        int mk[4];
        for (size_t k = 0; k < num_stages && k < 4; ++k) mk[k] = stage[num_stages - 1 - k]->m;
        return msresamp2_decim_lds((int)num_stages, mk, sizeof(T)) <= 64 * 1024;
    }
    int block_host(const T *x, size_t n, T *y) {
        if (n == 0) return YAGI_OK;
        const size_t nin = interp ? n : n * rate, nout = interp ? n * rate : n;
        YG_TRY(ws.x.ensure(nin * sizeof(T)));
        YG_TRY(ws.y.ensure(nout * sizeof(T)));
        YG_TRY(upload(ws.x.p, x, nin * sizeof(T), st));
        YG_TRY(block_dev(ws.x.template as<T>(), n, ws.y.template as<T>()));
        return download(y, ws.y.p, nout * sizeof(T), st);
    }
    float delay() const {                                            // :118-135
        float d = 0.0f;
        if (interp) {
            for (size_t i = 0; i < num_stages; ++i) { d *= 0.5f; d += (float)m_stage[num_stages - i - 1]; }
        } else {
            for (size_t i = 0; i < num_stages; ++i) { d *= 2.0f; d += 2.0f * (float)m_stage[i] - 1.0f; }
        }
        return d;
    }
};

}  // namespace yagi

#define YAGI_RESAMP2_IMPL(K, KT, T, C)                                                              \
    struct yagi_hip_resamp2_##K##_s : Resamp2Obj<KT> {};                                            \
    struct yagi_hip_msresamp2_##K##_s : MsResamp2Obj<KT> {};                                        \
    extern "C" {                                                                                    \
    int yagi_hip_resamp2_##K##_create(const float *hf, size_t m, float f0, yagi_hip_resamp2_##K *q) try { \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        CHECK_PTR(hf);                                                                              \
        auto o = std::make_unique<yagi_hip_resamp2_##K##_s>();                                      \
        YG_TRY(o->init(hf, m, f0));                                                                 \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_resamp2_##K##_create_kaiser(size_t m, float f0, float as_, yagi_hip_resamp2_##K *q) try { \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (m < 2) return fail(YAGI_ERR_CONFIG, "filter semi-length must be at least 2");           \
        if (f0 < -0.5f || f0 > 0.5f) return fail(YAGI_ERR_CONFIG, "f0 (%g) must be in [-0.5,0.5]", (double)f0); \
        if (as_ < 0.0f) return fail(YAGI_ERR_CONFIG, "as (%g) must be greater than zero", (double)as_); \
        if (m > (size_t)kR2MaxSemiLen) return fail(YAGI_ERR_CONFIG, "filter semi-length must not exceed %d", kR2MaxSemiLen); \
        std::vector<float> hf;                                                                      \
        YG_TRY(design_kaiser(4 * m + 1, 0.25f, as_ > 0.0f ? as_ : 1e-3f, 0.0f, hf));                \
        for (float &v : hf) v *= 0.5f;                 /* sinc(t/2) w(t) has centre 1: half-band = centre 1/2 */ \
        return yagi_hip_resamp2_##K##_create(hf.data(), m, f0, q);                                  \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_resamp2_##K##_destroy(yagi_hip_resamp2_##K q) try { delete q; return YAGI_OK; } catch (...) { return ::yagi::api_exception(); } \
    int yagi_hip_resamp2_##K##_clone(yagi_hip_resamp2_##K q, yagi_hip_resamp2_##K *out) try {       \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(out);                                                                             \
        *out = nullptr;                                                                             \
        auto o = std::make_unique<yagi_hip_resamp2_##K##_s>();                                      \
        YG_TRY(q->clone_into(*o));                                                                  \
        *out = o.release();                                                                         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_resamp2_##K##_reset(yagi_hip_resamp2_##K q) try { CHECK_Q(q); return q->reset(); } catch (...) { return ::yagi::api_exception(); } \
    int yagi_hip_resamp2_##K##_set_stream(yagi_hip_resamp2_##K q, yagi_stream_t s) try {            \
        CHECK_Q(q);                                                                                 \
        if (q->st == to_stream(s)) return YAGI_OK;                                                  \
        YG_HIP(hipStreamSynchronize(q->st));                                                        \
        q->st = to_stream(s);                                                                       \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_resamp2_##K##_set_scale(yagi_hip_resamp2_##K q, C scale) try { CHECK_Q(q); q->scale = scale; return YAGI_OK; } catch (...) { return ::yagi::api_exception(); } \
    int yagi_hip_resamp2_##K##_get_scale(yagi_hip_resamp2_##K q, C *scale) try {                    \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(scale);                                                                           \
        *scale = q->scale;                                                                          \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_resamp2_##K##_get_delay(yagi_hip_resamp2_##K q, size_t *delay) try {               \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(delay);                                                                           \
        *delay = 2 * (size_t)q->m - 1;                                                              \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_resamp2_##K##_execute_block(yagi_hip_resamp2_##K q, int mode, const T *x, size_t nx, T *y, size_t ny) try { \
        CHECK_Q(q);                                                                                 \
        YG_TRY(q->check_lengths(mode, nx, ny));                                                     \
        if (nx == 0) return YAGI_OK;                                                                \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        return q->block_host(mode, x, nx, y);                                                       \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_resamp2_##K##_execute_block_dev(yagi_hip_resamp2_##K q, int mode, const T *x, size_t nx, T *y, size_t ny) try { \
        CHECK_Q(q);                                                                                 \
        YG_TRY(q->check_lengths(mode, nx, ny));                                                     \
        if (nx == 0) return YAGI_OK;                                                                \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, nx, y, ny);                                                                \
        return q->block_dev(mode, x, nx, y);                                                        \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_msresamp2_##K##_create_taps(int interp, size_t num_stages, const size_t *m_stage,  \
                                             const float *hf_all, yagi_hip_msresamp2_##K *q) try {  \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        if (num_stages) { CHECK_PTR(m_stage); CHECK_PTR(hf_all); }                                  \
        auto o = std::make_unique<yagi_hip_msresamp2_##K##_s>();                                    \
        YG_TRY(o->init(interp != 0, num_stages, m_stage, hf_all));                                  \
        *q = o.release();                                                                           \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_msresamp2_##K##_create(int interp, size_t num_stages, float fc, float f0, float as_, \
                                        yagi_hip_msresamp2_##K *q) try {                            \
        CHECK_PTR(q);                                                                               \
        *q = nullptr;                                                                               \
        std::vector<size_t> ms;                                                                     \
        YG_TRY(msresamp2_stage_lengths(num_stages, fc, as_, ms));                                   \
        if (f0 != 0.0f) return fail(YAGI_ERR_CONFIG, "non-zero center frequency not yet supported"); \
        std::vector<float> all, hf;                                                                 \
        for (size_t i = 0; i < num_stages; ++i) {                                                   \
            YG_TRY(design_kaiser(4 * ms[i] + 1, 0.25f, as_ + 5.0f, 0.0f, hf));                      \
            for (float &v : hf) v *= 0.5f;                                                          \
            all.insert(all.end(), hf.begin(), hf.end());                                            \
        }                                                                                           \
        return yagi_hip_msresamp2_##K##_create_taps(interp, num_stages, ms.data(), all.data(), q);  \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_msresamp2_##K##_destroy(yagi_hip_msresamp2_##K q) try { delete q; return YAGI_OK; } catch (...) { return ::yagi::api_exception(); } \
    int yagi_hip_msresamp2_##K##_clone(yagi_hip_msresamp2_##K q, yagi_hip_msresamp2_##K *out) try { \
        CHECK_Q(q);                                                                                 \
        CHECK_PTR(out);                                                                             \
        *out = nullptr;                                                                             \
        auto o = std::make_unique<yagi_hip_msresamp2_##K##_s>();                                    \
        o->st = q->st;                                                                              \
        o->interp = q->interp;                                                                      \
        o->num_stages = q->num_stages;                                                              \
        o->rate = q->rate;                                                                          \
        o->m_stage = q->m_stage;                                                                    \
        for (auto &s : q->stage) {                                                                  \
            auto c = std::make_unique<Resamp2Obj<KT>>();                                            \
            YG_TRY(s->clone_into(*c));                                                              \
            o->stage.push_back(std::move(c));                                                       \
        }                                                                                           \
        *out = o.release();                                                                         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_msresamp2_##K##_reset(yagi_hip_msresamp2_##K q) try {                              \
        CHECK_Q(q);                                                                                 \
        for (auto &s : q->stage) YG_TRY(s->reset());                                                \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_msresamp2_##K##_set_stream(yagi_hip_msresamp2_##K q, yagi_stream_t s) try {        \
        CHECK_Q(q);                                                                                 \
        if (q->st == to_stream(s)) return YAGI_OK;                                                  \
        YG_HIP(hipStreamSynchronize(q->st));                                                        \
        q->st = to_stream(s);                                                                       \
        for (auto &g : q->stage) g->st = q->st;                                                     \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_msresamp2_##K##_get_params(yagi_hip_msresamp2_##K q, int *interp, size_t *num_stages, \
                                            float *delay, size_t *m_stage) try {                    \
        CHECK_Q(q);                                                                                 \
        if (interp) *interp = q->interp ? 1 : 0;                                                    \
        if (num_stages) *num_stages = q->num_stages;                                                \
        if (delay) *delay = q->delay();                                                             \
        if (m_stage) for (size_t i = 0; i < q->num_stages; ++i) m_stage[i] = q->m_stage[i];         \
        return YAGI_OK;                                                                             \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_msresamp2_##K##_execute_block(yagi_hip_msresamp2_##K q, const T *x, size_t nx, T *y, size_t ny) try { \
        CHECK_Q(q);                                                                                 \
        size_t n = 0;                                                                               \
        YG_TRY(q->check_lengths(nx, ny, &n));                                                       \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        return q->block_host(x, n, y);                                                              \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    int yagi_hip_msresamp2_##K##_execute_block_dev(yagi_hip_msresamp2_##K q, const T *x, size_t nx, T *y, size_t ny) try { \
        CHECK_Q(q);                                                                                 \
        size_t n = 0;                                                                               \
        YG_TRY(q->check_lengths(nx, ny, &n));                                                       \
        if (n == 0) return YAGI_OK;                                                                 \
        CHECK_PTR(x);                                                                               \
        CHECK_PTR(y);                                                                               \
        CHECK_NOALIAS(x, nx, y, ny);                                                                \
        return q->block_dev(x, n, y);                                                               \
    } catch (...) { return ::yagi::api_exception(); }                                               \
    }

YAGI_RESAMP2_IMPL(rrrf, RRRF, float, float)
YAGI_RESAMP2_IMPL(crcf, CRCF, yagi_cf32, float)
YAGI_RESAMP2_IMPL(cccf, CCCF, yagi_cf32, yagi_cf32)
