// kernels.hpp -- launcher declarations shared by the kernel translation units and capi.hip.
// Every launcher is asynchronous on `st`, performs no allocation and no host synchronisation
// (so a caller may capture it into a hipGraph), and returns a yagi_status.
#pragma once
#include "common.hpp"

namespace yagi {

// Type combinations of the FIR family (T = sample/output type, C = coefficient type).
struct RRRF { using T = float; using C = float; static constexpr int id = 0; };
struct CRCF { using T = cf32;  using C = float; static constexpr int id = 1; };
struct CCCF { using T = cf32;  using C = cf32;  static constexpr int id = 2; };

// ---- comm.cpp --------------------------------------------------------------------------------
}  // namespace yagi
// RCCL communicator handle of the C ABI (yagi_hip_comm): the ncclComm_t, this rank, a side stream for the collectives
// and the events that chain it to an object's stream
struct yagi_hip_comm_s {
    void *nccl = nullptr;
    int rank = 0, nranks = 1;
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> ev;
    hipEvent_t done = nullptr;
};
namespace yagi {
using Comm = yagi_hip_comm_s;
// all-gather `bytes_per_rank` bytes from every rank into recv (rank-major), asynchronous on st
int comm_all_gather(Comm *c, const void *send, void *recv, size_t bytes_per_rank, hipStream_t st);

// ---- misc_kernels.hip ----------------------------------------------------------------------
int launch_gen_real(uint64_t seed, uint64_t first, size_t n, float *x, hipStream_t st);
int launch_gen_complex(uint64_t seed, uint64_t first, size_t n, cf32 *x, hipStream_t st);

// y[0] = post * sum_i a[i] * b[rev ? n-1-i : i]; `partials` needs dotprod_num_partials(n)
// elements of the output type.  Fixed-order two-pass combine => bitwise reproducible.
size_t dotprod_num_partials(size_t n);
template <class A, class B, class O, class S>
int launch_dotprod(const A *a, const B *b, size_t n, bool rev_b, S post, O *partials, O *y,
                   hipStream_t st);

// new_win = last L samples of (old_win ++ x[0..n))
template <class T>
int launch_update_window(const T *old_win, const T *x, size_t n, int L, T *new_win, hipStream_t st);

// ---- fir_kernels.hip -----------------------------------------------------------------------
// y[i] = scale * sum_{k<L} h[k] * X[i*M - k],  X = win(L samples, oldest first) ++ x,
// X index 0 = x[0].  i in [0, ny).  taps in natural order h[0..L).
template <class K>
int launch_fir_block(const typename K::T *win, const typename K::T *x, const typename K::C *taps,
                     int L, int M, typename K::C scale, typename K::T *y, size_t ny, hipStream_t st,
                     size_t x_len = 0 /* samples readable at x; 0 = ny*M */,
                     typename K::T *win_next = nullptr /* if set: receives the window after the block (last L samples of
                                                          win ++ x[0..ny*M)), written by the kernel's last workgroup */);

// polyphase bank, all branches per input sample: y[n*nf + i] = scale * sum_k hb[i][k] X[n-k]
// branch taps hb laid out [nf][Ls] in natural (newest-first) order.
template <class K>
int launch_firpfb_all(const typename K::T *win, const typename K::T *x, const typename K::C *hb,
                      int nf, int Ls, typename K::C scale, typename K::T *y, size_t n, hipStream_t st,
                      typename K::T *win_next = nullptr /* see launch_fir_block */);
// branch chosen per sample: y[n] = scale * sum_k hb[idx[n]][k] X[n-k]
template <class K>
int launch_firpfb_select(const typename K::T *win, const typename K::T *x, const typename K::C *hb,
                         const uint32_t *idx, int nf, int Ls, typename K::C scale,
                         typename K::T *y, size_t n, hipStream_t st);

// Rresamp: y[blk*P + n] = scale * sum_k hb[(n*Q) % P][k] X[blk*Q + (n*Q)/P - k], n < P, blk < nblocks
template <class K>
int launch_rresamp(const typename K::T *win, const typename K::T *x, const typename K::C *hb, int P, int Q,
                   int Ls, typename K::C scale, typename K::T *y, size_t nblocks, hipStream_t st,
                   typename K::T *win_next = nullptr /* see launch_fir_block */);

// ---- stream_kernels.hip (crcf M=1 hot case; headline fused FIR -> 4096-pt FFT) -----------------
// taps_pad = h zero-padded to Lp = roundup(L, 32) floats.
constexpr int kSlideMaxTaps = 1024;
int launch_fir_crcf_slide(const cf32 *win, const cf32 *x, const float *taps_pad, int L, int Lp,
                          float scale, cf32 *y, size_t ny, hipStream_t st, cf32 *win_next = nullptr);
int launch_firfft_crcf_4096(const cf32 *win, const cf32 *x, const float *taps_pad, const float *apack,
                            int L, int Lp, int Lm, float scale, const cf32 *tw4096, cf32 *spectra,
                            size_t nframes, int variant, hipStream_t st);
// MFMA Toeplitz form (L <= 256): Lm = mfma_lp_for(L) in {64,128,256} (0 = unsupported);
// apack = pack_toeplitz_taps(h, L, Lm), toeplitz_pack_floats(Lm) floats.
int mfma_lp_for(int L);
size_t toeplitz_pack_floats(int Lp);
void pack_toeplitz_taps(const float *h, int L, int Lp, float *apack_host);
int launch_fir_crcf_mfma(const cf32 *win, const cf32 *x, const float *apack, int L, int Lm, float scale,
                         cf32 *y, size_t ny, hipStream_t st, cf32 *win_next = nullptr);

// Fast convolution (overlap-save, 4096-pt blocks) form of firfilt_crcf: hs = FFT_4096{[h;0]} (unscaled),
// forward table twf, backward table twb; 1 <= L <= 2049.
// x[-pre .. x_avail) must be readable (chunked processing of one long block); ny outputs are produced.
int launch_fir_crcf_fftconv(const cf32 *win, const cf32 *x, size_t pre, size_t x_avail, const cf32 *hs,
                            float scale, int L, const cf32 *twf, const cf32 *twb, cf32 *y, size_t ny,
                            hipStream_t st, cf32 *win_next = nullptr);
// win_next (optional): receives the filter window after the call, the last L samples of win ++ x[0, x_avail)
// same kernel for complex taps (hs = FFT of the complex taps, complex scale) and for real samples (two
// blocks per transform as its real and imaginary parts)
int launch_fir_cccf_fftconv(const cf32 *win, const cf32 *x, size_t pre, size_t x_avail, const cf32 *hs,
                            cf32 scale, int L, const cf32 *twf, const cf32 *twb, cf32 *y, size_t ny,
                            hipStream_t st, cf32 *win_next = nullptr);
int launch_fir_rrrf_fftconv(const float *win, const float *x, size_t pre, size_t x_avail, const cf32 *hs,
                            float scale, int L, const cf32 *twf, const cf32 *twb, float *y, size_t ny,
                            hipStream_t st, float *win_next = nullptr);

// frequency-domain form of the firfilt_crcf -> 4096-pt FFT stream (1 <= L <= 257): FFT{h}.FFT{x_f} + FFT{boundary
// correction}; hs_scaled = scale * FFT{h}; gfft_scaled = scale * conj(DFT_512{g}) / 512, g[j] = h[L-1-j]; tw_stream = the
// table of make_stream_twiddles (capi.hip); win_next <- last L samples of x.
int launch_firfft_crcf_4096_freq(const cf32 *win, const cf32 *x, const cf32 *hs_scaled, const cf32 *gfft_scaled,
                                 int L, const cf32 *tw_stream, cf32 *spectra, cf32 *win_next, size_t nframes,
                                 hipStream_t st);

int launch_scale_cf32(const cf32 *src, float s, cf32 *dst, size_t n, hipStream_t st);

// ---- resamp2_kernels.hip -------------------------------------------------------------------
// forms of Resamp2 (resamp2.rs:104-180); values are the `mode` argument of the C ABI
enum { kR2Filter = 0, kR2Analyzer = 1, kR2Synthesizer = 2, kR2Decim = 3, kR2Interp = 4 };
constexpr int kR2MaxSemiLen = 1024;
// MsResamp2 decimator chain in one launch (resamp2_kernels.hip): stages in processing order (full rate first)
template <class T, class C>
int launch_msresamp2_decim(int ns, const int *m, const C *scale, const C *const *h1, const T *const *state,
                           T *const *state_next, const T *x, T *y, size_t nout, hipStream_t st);
size_t msresamp2_decim_lds(int ns, const int *m, size_t elem);
// MsResamp2 interpolator chain in one launch: stages in processing order (input rate first)
template <class T, class C>
int launch_msresamp2_interp(int ns, const int *m, const C *scale, const C *const *h1, const T *const *state,
                            T *const *state_next, const T *x, T *y, size_t nin, hipStream_t st);
size_t msresamp2_interp_lds(int ns, const int *m, size_t elem);
// one block of nx input samples; state = [w0 (2m, oldest first)][w1 (2m)]; state_next receives the windows after the
// block (must not alias state).  Outputs: filter 2 nx ((y0,y1) pairs), analyzer / synthesizer nx, decim nx/2, interp 2 nx.
template <class T, class C>
int launch_resamp2(int mode, const T *state, const T *x, size_t nx, const C *h1, int m, C scale, int toggle, T *y,
                   T *state_next, hipStream_t st);

// ---- fft_kernels.hip -----------------------------------------------------------------------
struct FftPlanDev {
    int n = 0;
    int dir = 0;
    int nfac = 0;
    int fac[16] = {0};
    const cf32 *tw = nullptr;        // W_n^m, m in [0,n), sign per direction
    // Bluestein (chirp-z) form for sizes with a large prime factor: X = w . IFFT_m(FFT_m(x . w) . FFT_m(b)) / m
    int bs_m = 0;                    // 0 = not used; else the power-of-two convolution length >= 2n-1
    const cf32 *bs_w = nullptr;      // chirp w[k] = e^{-+ j pi k^2 / n}, k < n
    const cf32 *bs_bf = nullptr;     // FFT_m of the circular chirp filter conj(w[|k|])
    const cf32 *bs_twf = nullptr;    // W_m tables, forward / backward (8192: followed by the W_4096 table)
    const cf32 *bs_twb = nullptr;
    cf32 *bs_scratch = nullptr;      // 2 * bs_chunk * m points
    int bs_chunk = 0;                // transforms per pass through the scratch
    const FftPlanDev *bs_fwd = nullptr, *bs_bwd = nullptr;   // host pointers: the m-point plans (own resources when m > 8192)
    // four-step form above 8192 points: n = n1 * n2 (both <= 8192), column transforms, twiddle, row transforms
    int fs_n1 = 0, fs_n2 = 0;        // 0 = not used
    const FftPlanDev *fs_p1 = nullptr, *fs_p2 = nullptr;      // host pointers: the n1- and n2-point plans
    cf32 *fs_scratch = nullptr;      // 2 * fs_chunk * n points
    int fs_chunk = 0;
    const cf32 *fs_wn = nullptr;     // W_n table when both factors are powers of two <= 256: two-launch form, no transposes
    const cf32 *fs_wlo4 = nullptr, *fs_whi4 = nullptr; // four-step form: the same split tables for its twiddle transposition
    const cf32 *fs_wlo = nullptr, *fs_whi = nullptr;   // n = 256 n2 >= 2^16 (fft_tile256_kernel): W_n^j, j < 4096, and W_n^{4096 j}
};
constexpr int kFftMaxLds = 8192;     // complex points held in LDS by the one-kernel path
constexpr int kFftTwoPassMixedMax = 1024;  // n = n1 n2 with both factors up to this: two launches (fft_mixed_twopass_kernel)
constexpr size_t kFftMaxPow2 = (size_t)1 << 24;    // largest power of two (four-step); any other n up to 2^23 (Bluestein)
int launch_fft_batch(const FftPlanDev &p, const cf32 *in, cf32 *out, size_t batch, hipStream_t st);
int launch_fft_shift(cf32 *buf, size_t n, size_t batch, hipStream_t st);

// ---- fftfilt_kernels.hip (FftFilt, src/filter/fftfilt.rs:103-138) ------------------------------------
template <class T>
int launch_fftfilt_pad(const T *x, int n, size_t nblocks, cf32 *time, hipStream_t st);
int launch_fftfilt_mul(cf32 *freq, const cf32 *hf, int n2, size_t nblocks, hipStream_t st);
template <class T, class C>
int launch_fftfilt_ola(const cf32 *t, const cf32 *w, int n, size_t nblocks, C scale, T *y, cf32 *w_next,
                       hipStream_t st);

// ---- spgram_kernels.hip (fft::Spgram, src/fft/spgram.rs:237-316) ---------------------------------------
template <class T>
int launch_spgram_frames(const T *win, const T *x, const float *w, int wlen, int nfft, long long first,
                         int delay, size_t nframes, cf32 *time, hipStream_t st);
size_t spgram_accum_scratch_floats(int nfft, size_t nframes);
// `part`: spgram_accum_scratch_floats(nfft, nframes) floats of scratch
int launch_spgram_accum(const cf32 *freq, int nfft, size_t nframes, float alpha, float gamma, bool first_ever,
                        float *psd, float *part, hipStream_t st);
// fused taper -> FFT -> |X|^2 -> weighted accumulation (nfft in {256,512,1024,2048,4096});
// part: spgram_fused_scratch_floats(nfft, nframes) floats; twn = the plan's W_nfft table
bool spgram_fused_supported(int nfft);
size_t spgram_fused_scratch_floats(int nfft, size_t nframes);
template <class T>
int launch_spgram_fused(int nfft, const T *win, const T *x, size_t x_len, const float *w, int wlen, long long first,
                        int delay, size_t nframes, float alpha, float gamma, bool first_ever, const cf32 *twn,
                        float *psd, float *part, hipStream_t st);
int launch_spgram_psd(const float *psd, int nfft, float scale, bool in_db, float *out, hipStream_t st);

// ---- chan_kernels.hip ----------------------------------------------------------------------
// firpfbch analyzer: hist = the (p-1)*M samples preceding x[0] (oldest first).
// hist_next (optional): the object's other history buffer; when the kernel taken can write the history after the
// block itself, *hist_written is set and the caller skips its own update.
int launch_firpfbch(const cf32 *hist, const cf32 *x, const float *h, int M, int p,
                    const cf32 *twM, cf32 *y, size_t nframes, hipStream_t st, cf32 *hist_next = nullptr,
                    bool *hist_written = nullptr);
// firpfbch synthesizer: x = nframes frames of M channel samples, hist = the (p-1)*M channel samples before them;
// y[f*M + i] = sum_n h[i + n*M] * IDFT_M(frame f-n)[i]
int launch_firpfbch_syn(const cf32 *hist, const cf32 *x, const float *h, int M, int p,
                        const cf32 *twM, cf32 *y, size_t nframes, hipStream_t st);
// firpfbch2 synthesizer: x = nsteps steps of M channel samples, hist = the (4m-1)*M channel samples before them,
// step0 = index of the first step (parity selects the half the outputs come from); y = nsteps*M/2 samples
int launch_firpfbch2_syn(const cf32 *hist, int hist_len, const cf32 *x, const float *h, int M, int m,
                         const cf32 *twM, uint64_t step0, cf32 *y, size_t nsteps, hipStream_t st);
// firpfbch2 analyzer: hist = the 2*m*M - M/2 ... samples preceding x[0]; step0 = index of the
// first step (parity selects the half rotation).  rank/nranks select sub-bands k = rank + nranks*q.
int launch_firpfbch2(const cf32 *hist, int hist_len, const cf32 *x, const float *h, int M, int m,
                     const cf32 *twM, uint64_t step0, int rank, int nranks, cf32 *y, size_t nsteps,
                     hipStream_t st, cf32 *hist_next = nullptr, bool *hist_written = nullptr);
int launch_firpfbch2_assemble(const cf32 *gathered, size_t nsteps, int M, int nranks, cf32 *y,
                              hipStream_t st);

}  // namespace yagi
