// freq_kernels.hip -- the headline stream kernel: firfilt_crcf (<= 257 taps) -> 4096-point forward FFT per frame,
// filtered in the frequency domain (reference composition: src/filter/fir/firfilt.rs:267-278 feeding
// src/fft/mod.rs:45-48).  One launch per block, 16 B of HBM traffic per input sample.
#include "fft_core.hpp"
#include "kernels.hpp"

namespace yagi {

// ---------------------------------------------------------------------------------------------
// Frame f of the filtered stream is y_f[n] = sum_k h[k] X[4096 f + n - k] (n < 4096, k < L, X = stream).
// Splitting off the terms that reach back into the previous frame:
//     y_f = h (*) x_f  +  c_f              ((*) = 4096-point circular convolution, x_f = frame f of X)
//     c_f[n] = sum_{k > n} h[k] (X[4096 f + n - k] - x_f[4096 + n - k]),   n < L-1,  zero elsewhere
// so, the FFT being linear,
//     FFT{y_f} = FFT{h} . FFT{x_f} + FFT{c_f}.
// The workgroup reads its frame once, transforms it, multiplies by FFT{h} (precomputed, scale folded in), and adds
// the transform of the (L-1)-sample correction -- whose first butterfly pass degenerates because only its first 256
// samples are non-zero.  No inverse transform, no intermediate stream: 1 2/3 FFTs per frame against 3.1 for
// overlap-save + FFT, and half the HBM traffic.
//
// c_f is a triangular Toeplitz product.  With d[m] = X[4096 f - (L-1) + m] - x_f[4096 - (L-1) + m] (m < L-1, zero
// beyond) and g[j] = scale h[L-1-j] (j < L-1, zero beyond):  c_f[n] = sum_j g[j] d[n + j], only n + j < L-1 non-zero
// -- a correlation of two sequences of at most 256 samples, evaluated as ONE 512-point circular correlation by wave 0
// (FAST CORRECTION SUM below; rounds 1-2 summed the triangle directly: 160 packed FMAs on each of 238 lanes).
//
// TWIDDLES.  Three of the four twiddle sets of a frame are powers of W_256 (pass 2 of both transforms and the input
// rotation of the correction transform): they come from a 16 x 16 table T[b][c] = W_256^{bc} in LDS (row pitch 17:
// the 16 rows a wave instruction touches sit on 16 distinct bank pairs), as immediate-offset ds_reads -- no address
// arithmetic, no global gathers (15 gathers per pass had been the L1/TA bottleneck, 4 gathers + 11 products cost
// 22 packed instructions per set).  The pass-1 set W_4096^{tc} keeps "four exact entries + products"; its four
// entries W^{t 2^k} and the correction transform's W^{b'c} are rows of an auxiliary table indexed by the lane:
// coalesced loads.
//
// LOADS AND STORES.  The frame's 32 eight-byte accesses per lane are buffer accesses: one VGPR byte offset, 2 KiB
// steps in SGPR offsets (the flat form spent ~3 VALU + a carry-hazard nop on each).  Everything the head needs -- the
// table row, wave 0's correction operands, the 16 frame loads -- is requested in one go before the first LDS write:
// one memory round trip; wave 0's operands go out AHEAD of its frame loads (vmcnt retires in order), all of them plain
// loads whose waits the compiler counts itself (no hand-counted s_waitcnt is left in this kernel).  The frame is read
// once and the spectra are written once, so the frame loads bypass the CU's L1 (sc1: the tables the four resident
// workgroups share stay there) and the stores are non-temporal (nt): interleaved A/B on the streamed workload,
// tools/ab_libs.py: stores nt -4.6 ... -6.7 %, loads sc1 a further -2.3 %; nt LOADS +10 %, sc0 on either +13 ... +18 %.
//
// WHAT BOUNDS IT (profiles/r03_notes.md).  The chip's power envelope: tools/kb_power.py reads 1383 W (cap 1400 W) and a
// shader clock throttled to 2.05 GHz while this kernel streams random data, 1225 W at 2.39 GHz on all-zero input --
// the same instruction stream and traffic then run 15 % faster (tools/kb_zero.py).  No unit is saturated and the
// dependent chain of a frame is not the limit either (a loop form that prefetches the next frame and shortens the
// chain by 8 % runs no faster): energy per frame is what a faster version has to lower.
//
// Tried and dropped over the rounds, all correct, all slower or equal (numbers in profiles/r0{1,2,3}_notes.md): a
// persistent grid with the next frame's loads prefetched (static split, ticket counter, and -- round 3 -- a loop form
// under the pipelined block calls that hide its uneven finish), the corrections of a block as a launch of their own,
// FFT{h} / twiddle loads hoisted to the kernel head, the correction transform's two output factors as one gathered
// table entry, a two-wave workgroup per frame, the triangle on the matrix pipe, barriers sunk below register-only
// work, an L2 prefetch of a later frame, s_setprio over the head.
// ---------------------------------------------------------------------------------------------
constexpr int kFreqMaxTaps = 257;
constexpr unsigned kOffCvs = kFft4096LdsFloat2;          // c_f, 256 float2 (behind the exchange buffer)
constexpr unsigned kOffT256 = kOffCvs + 256;             // T[b][c] = W_256^{bc}, 16 rows of 17 float2
constexpr unsigned kFreqLdsFloat2 = kOffT256 + 16 * kTRow;   // 39 040 B -> 4 workgroups per CU

// passes 2 and 3 of the frame transform (exchange-1 data stored, a barrier ago), pass-2 twiddles W_256^{b'c'} = T[b'][c']
__device__ __forceinline__ void freq_pass23(float2 (&v)[16], float2 *__restrict__ lds, const float2 *__restrict__ T) {
    const unsigned t = threadIdx.x;
    {
        const unsigned c = t >> 4, bp = t & 15;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = lds[c * kEx1Stride + 16 * a + bp];
        __syncthreads();
        dft16<-1>(v);
#pragma unroll
        for (int cp = 0; cp < 16; ++cp) {
            float2 u = v[dft16_pos(cp)];
            if (cp) u = cmul(u, T[bp * kTRow + cp]);
            lds[bp * kEx2Stride + cp * 16 + c] = u;
        }
    }
    __syncthreads();
    float2 w[16];
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) w[bp] = lds[bp * kEx2Stride + t];
    dft16<-1>(w);
#pragma unroll
    for (int d = 0; d < 16; ++d) v[d] = w[dft16_pos(d)];
    __syncthreads();
}

// Transform of a sequence that is zero from sample 256 on.  The first 16-point butterfly degenerates to
// Z_c[b] = W4096^{bc} x[b], so pass 1 and its whole exchange are skipped: the 256 samples xs[0..256) sit in LDS outside
// the exchange buffer `lds`, and the pass-2 lane (c, b') rebuilds its inputs
//     Z_c[16a + b'] = x[16a + b'] (W^{16c})^a W^{b'c}            ((W^{16c})^a = T[c][a])
// the factor W^{b'c} (`wbc`), common to all 16 inputs, is applied to the butterfly's outputs together with the pass-2
// twiddle.  On return lane t holds X[t + 256 d] in v[d].  One barrier; `lds` must be free on entry.
__device__ __forceinline__ void freq_head256(const float2 *__restrict__ xs, float2 (&v)[16], float2 *__restrict__ lds,
                                             const float2 *__restrict__ T, float2 wbc) {
    const unsigned t = threadIdx.x;
    const unsigned c = t >> 4, bp = t & 15;
    v[0] = xs[bp];
#pragma unroll
    for (int a = 1; a < 16; ++a) v[a] = cmul(xs[16 * a + bp], T[c * kTRow + a]);
    dft16<-1>(v);
#pragma unroll
    for (int cp = 0; cp < 16; ++cp) {
        float2 u = cmul(v[dft16_pos(cp)], wbc);
        if (cp) u = cmul(u, T[bp * kTRow + cp]);
        lds[bp * kEx2Stride + cp * 16 + c] = u;
    }
    __syncthreads();
    float2 w[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) w[b] = lds[b * kEx2Stride + t];
    dft16<-1>(w);
#pragma unroll
    for (int d = 0; d < 16; ++d) v[d] = w[dft16_pos(d)];
}

// ---------------------------------------------------------------------------------------------
// FAST CORRECTION SUM.  c_f[n] = sum_j g[j] d[n + j] is a correlation of two sequences of at most 256 samples, so
// with N = 512 >= 255 + 256 it is one circular correlation:  c_f = IDFT_512{ DFT_512{d} . conj(DFT_512{g}) }.
// conj(DFT{g}) / 512 is a table (host, f64); the two 512-point transforms run in ONE wave (8 points per lane,
// 8 x 8 x 8, in-wave exchanges through the idle exchange buffer: no workgroup barrier), first pass pruned to the
// 256 non-zero inputs, last pass pruned to the 256 outputs that are kept.  Against the balanced triangle
// (160 packed FMAs + 32 16-byte LDS reads on each of 238 lanes, three barriers) this is ~270 packed operations and
// 64 8-byte LDS accesses on 64 lanes and one barrier: a sixth fewer vector instructions and a fifth less LDS traffic
// per frame -- and the kernel is limited by the chip's power envelope, not by a unit (tools/kb_zero.py: the same
// launches on all-zero input run 15 % faster), so energy per frame is what counts.  Wave 0 does it while the other
// three wait for their frame loads; its operands (the 2 x 4 correction loads, six twiddles, eight table entries)
// are requested ahead of its frame loads, as plain loads the compiler counts itself.
// ---------------------------------------------------------------------------------------------
template <int SIGN>
__device__ __forceinline__ float2 w8(float c, float s) { return make_float2(c, SIGN < 0 ? -s : s); }
// X[k] = E[k & 3] + W_8^k O[k & 3]; E / O = 4-point transforms of the even / odd inputs (natural order)
template <int SIGN, int NOUT>
__device__ __forceinline__ void dft8_combine(const float2 (&E)[4], float2 (&O)[4], float2 (&X)[8]) {
    const float r2 = 0.70710678118654752f;
    O[1] = cmul_k(O[1], w8<SIGN>(r2, r2));
    O[3] = cmul_k(O[3], w8<SIGN>(-r2, r2));
    X[0] = cadd(E[0], O[0]);
    X[1] = cadd(E[1], O[1]);
    X[3] = cadd(E[3], O[3]);
    if (NOUT > 4) {
        X[4] = csub(E[0], O[0]);
        X[5] = csub(E[1], O[1]);
        X[7] = csub(E[3], O[3]);
        addsub_rot<SIGN>(E[2], O[2], X[2], X[6]);
    } else {
        float2 unused;
        addsub_rot<SIGN>(E[2], O[2], X[2], unused);
    }
}
// 8-point transform, natural order in and out; NIN = 4: inputs 4..7 are zero; NOUT = 4: only outputs 0..3 are made
template <int SIGN, int NIN = 8, int NOUT = 8>
__device__ __forceinline__ void dft8(const float2 (&v)[8], float2 (&X)[8]) {
    float2 E[4], O[4];
    if (NIN == 4) {
        E[0] = cadd(v[0], v[2]); E[2] = csub(v[0], v[2]); addsub_rot<SIGN>(v[0], v[2], E[1], E[3]);
        O[0] = cadd(v[1], v[3]); O[2] = csub(v[1], v[3]); addsub_rot<SIGN>(v[1], v[3], O[1], O[3]);
    } else {
        E[0] = v[0]; E[1] = v[2]; E[2] = v[4]; E[3] = v[6];
        O[0] = v[1]; O[1] = v[3]; O[2] = v[5]; O[3] = v[7];
        dft4<SIGN>(E[0], E[1], E[2], E[3]);
        dft4<SIGN>(O[0], O[1], O[2], O[3]);
    }
    dft8_combine<SIGN, NOUT>(E, O, X);
}
constexpr unsigned kW512S1 = 72, kW512S2 = 68, kW512Ex2 = 8 * kW512S1;   // in-wave exchange layouts (conflict-free)
// One 512-point transform inside a wave: lane l holds v[a] = x[64 a + l]; on return X[d] = X_512[l + 64 d].
// w1, w2, w4 = W_512^{l, 2l, 4l}; u1, u2, u4 = W_64^{b', 2b', 4b'} (b' = l & 7), forward values (conjugated here for SIGN > 0).
template <int SIGN, int NIN, int NOUT>
__device__ __forceinline__ void wave_fft512(const float2 (&v)[8], float2 (&X)[8], float2 *__restrict__ scr, unsigned l,
                                            float2 w1, float2 w2, float2 w4, float2 u1, float2 u2, float2 u4) {
    float2 z[8], r[8];
    dft8<SIGN, NIN, 8>(v, z);
    {
        const float2 w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
        const float2 w[8] = {w1, w1, w2, w3, w4, w5, w6, w7};
        scr[l] = z[0];
#pragma unroll
        for (int c = 1; c < 8; ++c) scr[c * kW512S1 + l] = cmul_dir<SIGN>(z[c], w[c]);
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned c = l >> 3, bp = l & 7u;
#pragma unroll
    for (int a = 0; a < 8; ++a) r[a] = scr[c * kW512S1 + 8 * a + bp];
    dft8<SIGN, 8, 8>(r, z);
    {
        const float2 u3 = cmul(u1, u2), u5 = cmul(u4, u1), u6 = cmul(u4, u2), u7 = cmul(u4, u3);
        const float2 u[8] = {u1, u1, u2, u3, u4, u5, u6, u7};
        scr[kW512Ex2 + bp * kW512S2 + c] = z[0];
#pragma unroll
        for (int cp = 1; cp < 8; ++cp) scr[kW512Ex2 + bp * kW512S2 + 8 * cp + c] = cmul_dir<SIGN>(z[cp], u[cp]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < 8; ++b) r[b] = scr[kW512Ex2 + b * kW512S2 + l];
    __builtin_amdgcn_wave_barrier();                 // the next transform's writes stay behind these reads
    dft8<SIGN, 8, NOUT>(r, X);
}

// Wave 0's operands for the fast correction sum, requested ahead of its frame loads
struct CorrOperands {
    float2 dp[4], dq[4];     // X[4096 f - Lc + 64 a + l], x_f[4096 - Lc + 64 a + l]
    float2 w1, w2, w4, u1, u2, u4;
    float2 gc[8];            // conj(DFT_512{g})[l + 64 d] * scale / 512
};
__device__ __forceinline__ void freq_load_corr_fft(CorrOperands &o, const float2 *__restrict__ win,
                                                   const float2 *__restrict__ x, int L, unsigned f,
                                                   const float2 *__restrict__ tw, const float2 *__restrict__ gfft) {
    const unsigned l = threadIdx.x & 63u;
    const int Lc = L - 1;
    const float2 *xf = x + (size_t)f * 4096;
    const float2 *prev = Lc ? (f ? xf - Lc : win + 1) : xf;
    const float2 *own = xf + (4096 - (Lc ? Lc : 1));
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const unsigned m = 64u * a + l, mi = (int)m < Lc ? m : 0u;
        o.dp[a] = prev[mi];
        o.dq[a] = own[mi];
    }
    o.w1 = tw[8u * l]; o.w2 = tw[16u * l]; o.w4 = tw[32u * l];
    const unsigned bp = l & 7u;
    o.u1 = tw[64u * bp]; o.u2 = tw[128u * bp]; o.u4 = tw[256u * bp];
#pragma unroll
    for (int d = 0; d < 8; ++d) o.gc[d] = gfft[l + 64u * d];
}
// c_f[0 .. 256) -> cvs, by wave 0 (l = threadIdx.x < 64); scr = 1120 float2 of LDS nobody else touches meanwhile
__device__ __forceinline__ void freq_correction_fft(const CorrOperands &o, int L, float2 *__restrict__ scr,
                                                    float2 *__restrict__ cvs) {
    const unsigned l = threadIdx.x & 63u;
    const int Lc = L - 1;
    float2 v[8], D[8];
#pragma unroll
    for (int a = 0; a < 4; ++a) v[a] = (int)(64u * a + l) < Lc ? csub(o.dp[a], o.dq[a]) : make_float2(0.f, 0.f);
#pragma unroll
    for (int a = 4; a < 8; ++a) v[a] = make_float2(0.f, 0.f);
    wave_fft512<-1, 4, 8>(v, D, scr, l, o.w1, o.w2, o.w4, o.u1, o.u2, o.u4);
#pragma unroll
    for (int d = 0; d < 8; ++d) v[d] = cmul(D[d], o.gc[d]);
    wave_fft512<1, 8, 4>(v, D, scr, l, o.w1, o.w2, o.w4, o.u1, o.u2, o.u4);
#pragma unroll
    for (int d = 0; d < 4; ++d) cvs[l + 64u * d] = (int)(64u * d + l) < Lc ? D[d] : make_float2(0.f, 0.f);
}

// The headline kernel.  hs = scale * FFT_4096{[h; 0]}; gfft = scale * conj(DFT_512{g}) / 512; tw = the stream twiddle
// table (capi.hip make_stream_twiddles: W_4096^m, then rows W^{t 2^k} (k < 4), W^{(t&15)(t>>4)}, W_256^{(t&15)(t>>4)});
// win = the L-sample filter window before the call, win_next receives the one after it (last L samples of x; null =
// the caller produces it).
__global__ void __launch_bounds__(256, 4)
firfft_crcf_4096_freq_kernel(const float2 *__restrict__ win, const float2 *__restrict__ x,
                              const float2 *__restrict__ hs, const float2 *__restrict__ gfft, int L,
                              const float2 *__restrict__ tw, float2 *__restrict__ out,
                              float2 *__restrict__ win_next, unsigned nframes) {
    __shared__ __attribute__((aligned(16))) float2 lds[kFreqLdsFloat2];
    float2 *cvs = lds + kOffCvs;
    const float2 *T = lds + kOffT256;
    const float2 *ax = tw + 4096;
    const unsigned t = threadIdx.x, f = blockIdx.x;
    const bool wave0 = __builtin_amdgcn_readfirstlane(t >> 6) == 0;
    // ---- head: every load the frame needs, wave 0's correction operands first ----
    const float2 tw_t = ax[1280 + t];
    float2 v[16];
    CorrOperands co;
    if (wave0) freq_load_corr_fft(co, win, x, L, f, tw, gfft);
    __builtin_amdgcn_sched_barrier(0);
    {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (size_t)f * 4096, 32768u);
#pragma unroll
        for (unsigned a = 0; a < 16; ++a) v[a] = buf_ld(rx, 8u * t, 2048u * a);
    }
    __builtin_amdgcn_sched_barrier(0);
    lds[kOffT256 + (t >> 4) * kTRow + (t & 15u)] = tw_t;
    if (f == nframes - 1 && win_next) {
        const float2 *xf = x + (size_t)f * 4096;
        for (int i = t; i < L; i += 256) win_next[i] = xf[4096 - L + i];
    }
    if (wave0) freq_correction_fft(co, L, lds, cvs);
    __syncthreads();                                  // c_f and T published; the exchange buffer is free
    // ---- frame transform ----
    dft16<-1>(v);
    {
        float2 w[16];
        twiddle_powers_from(w, ax[t], ax[256 + t], ax[512 + t], ax[768 + t]);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float2 z = v[dft16_pos(c)];
            if (c) z = cmul(z, w[c]);
            lds[c * kEx1Stride + t] = z;
        }
    }
    __syncthreads();
    freq_pass23(v, lds, T);
    {
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(hs, 32768u);
        float2 hv[16];
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) hv[d] = buf_ld_aux<0>(rh, 8u * t, 2048u * d);      // a table: cached
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) v[d] = cmul(v[d], hv[d]);
    }
    // ---- correction transform, sum, store ----
    float2 u[16];
    freq_head256(cvs, u, lds, T, ax[1024 + t]);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)f * 4096, 32768u);
#pragma unroll
    for (unsigned d = 0; d < 16; ++d)
        buf_st(ro, 8u * t, 2048u * d, make_float2(v[d].x + u[d].x, v[d].y + u[d].y));
}

__global__ void scale_cf32_kernel(const float2 *__restrict__ src, float s, float2 *__restrict__ dst, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = make_float2(src[i].x * s, src[i].y * s);
}
// dst[i] = s * src[i] (the scaled copy of FFT{h} the stream kernel multiplies by)
int launch_scale_cf32(const cf32 *src, float s, cf32 *dst, size_t n, hipStream_t st) {
    if (n == 0) return YAGI_OK;
    scale_cf32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(reinterpret_cast<const float2 *>(src), s,
                                                                    reinterpret_cast<float2 *>(dst), (unsigned)n);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

int launch_firfft_crcf_4096_freq(const cf32 *win, const cf32 *x, const cf32 *hs_scaled, const cf32 *gfft_scaled,
                                 int L, const cf32 *tw_stream, cf32 *spectra, cf32 *win_next, size_t nframes,
                                 hipStream_t st) {
    if (nframes == 0) return YAGI_OK;
    if (L < 1 || L > kFreqMaxTaps)
        return fail(YAGI_ERR_CONFIG, "frequency-domain stream kernel needs 1..%d taps (got %d)", kFreqMaxTaps, L);
    if (nframes > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    firfft_crcf_4096_freq_kernel<<<(unsigned)nframes, 256, 0, st>>>(
        reinterpret_cast<const float2 *>(win), reinterpret_cast<const float2 *>(x),
        reinterpret_cast<const float2 *>(hs_scaled), reinterpret_cast<const float2 *>(gfft_scaled), L,
        reinterpret_cast<const float2 *>(tw_stream), reinterpret_cast<float2 *>(spectra),
        reinterpret_cast<float2 *>(win_next), (unsigned)nframes);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

}  // namespace yagi
