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
// beyond) and g[j] = scale h[L-1-j] (j < L-1, zero beyond):  c_f[n] = sum_j g[j] d[n + j], only n + j < L-1 non-zero.
//
// BALANCED TRIANGLE.  The triangle is cut into 238 work items (q, c): the 4 consecutive outputs n = 4q .. 4q+3
// against the 40 taps j = 40c .. 40c+39, for every c with 40c < 256 - 4q (64, 54, 44, 34, 24, 14, 4 output groups
// for c = 0 .. 6).  Lane t takes item t (items ordered chunk-major, so neighbouring lanes hold neighbouring output
// groups and keep the conflict-free 48-byte lane stride of the sample reads: 16 B of padding after every 4 samples);
// every lane runs the same 10 x 16 packed FMAs -- 160 per lane where a wave-per-tap-range split of the 256 x 256
// square costs 256.  The samples slide through registers (one 16-byte pair of LDS reads per 16 FMAs); taps are
// per-lane, read from LDS four at a time, the packed FMA broadcasts the half it needs.  Partial sums meet in LDS;
// output n adds its ceil((64 - n/4) / 10) partials in chunk order (fixed order: bitwise reproducible).
//
// TWIDDLES.  Three of the four twiddle sets of a frame are powers of W_256 (pass 2 of both transforms and the input
// rotation of the correction transform): they come from a 16 x 16 table T[b][c] = W_256^{bc} in LDS (row pitch 17:
// the 16 rows a wave instruction touches sit on 16 distinct bank pairs), as immediate-offset ds_reads -- no address
// arithmetic, no global gathers (15 gathers per pass had been the L1/TA bottleneck, 4 gathers + 11 products cost
// 22 packed instructions per set).  The pass-1 set W_4096^{tc} keeps "four exact entries + products"; its four
// entries W^{t 2^k} and the correction transform's W^{b'c} are rows of an auxiliary table indexed by the lane:
// coalesced loads.
//
// LOADS.  The frame's 48 eight-byte accesses per lane are buffer accesses: one VGPR byte offset, 2 KiB steps in SGPR
// offsets (the flat form spent ~3 VALU + a carry-hazard nop on each).  Everything the head needs -- taps, table
// row, the two correction loads, the 16 frame loads -- is requested in one go before the first LDS write: one
// memory round trip, and the correction loads are issued AHEAD of the frame loads as untracked asm loads (see
// freq_load_corr) so the correction arithmetic starts while the frame is still in flight.
//
// Measured on the 2^28-sample stream (profiles/r02_notes.md): 70-72 us per 2^24 samples for the round-1 form of
// this kernel -> 60-63 us.  Tried on top and dropped, all correct, all slower: a persistent grid with the next
// frame's loads prefetched (static split 68 us, ticket counter 81 us: one counter word saturates at ~90 tickets
// per us), the corrections of a block as a launch of their own (15 + 58 us), FFT{h} / twiddle loads hoisted to
// the kernel head (no change; with FFT{h} in registers from the start: spills), the correction transform's two
// output factors as one gathered table entry (+12 %: the gathers again), and a two-wave workgroup per frame (main wave:
// 64 x 64 transform of the frame in one wave's registers; correction wave: sum + transform of c_f; spectra added through
// LDS; two barriers per frame): 78.6 us -- at 2 waves per SIMD the long dependent chains are not hidden.
// ---------------------------------------------------------------------------------------------
constexpr int kFreqMaxTaps = 257;
constexpr unsigned kOffCvs = kFft4096LdsFloat2;          // c_f, 256 float2 (behind the exchange buffer)
constexpr unsigned kOffT256 = kOffCvs + 256;             // T[b][c] = W_256^{bc}, 16 rows of 17 float2
constexpr unsigned kTRow = 17;
constexpr unsigned kOffG = kOffT256 + 16 * kTRow;        // scaled taps g[0..288), 144 float2
constexpr unsigned kFreqLdsFloat2 = kOffG + 144;         // 40 064 B -> 4 workgroups per CU
constexpr unsigned kOffPart = 768;                       // partial sums behind d (inside the exchange buffer), 6 float2 per item
constexpr int kTriS = 40;                                // taps per work item

// sample m of d at float2 index m + 2 (m >> 2): 16 B of padding after every 4 samples
__device__ __forceinline__ unsigned pidx4(unsigned m) { return m + ((m >> 2) << 1); }
// first item of chunk c: sum_{c' < c} (64 - 10 c')
__device__ __forceinline__ constexpr unsigned tri_base(unsigned c) { return 64u * c - 5u * c * (c - 1u); }

// The two short loads of the frame-boundary correction: the L-1 samples before the frame (previous frame, or the
// filter window: win[L-k] = X[-k]) and the frame's own last L-1 samples; lanes >= L-1 read a valid dummy.
// They are ISSUED here as untracked asm loads and completed by freq_wait_corr: written as plain loads hipcc sinks
// them below the 16 frame loads into the `t < L-1` branch of their use, and the correction then waits for the whole
// frame (vmcnt retires in order).  Because they are older than every load the compiler tracks, its own vmcnt
// counts stay sufficient; freq_wait_corr's count = the younger loads that may stay in flight behind them.
__device__ __forceinline__ void freq_load_corr(const float2 *__restrict__ win, const float2 *__restrict__ x, int L,
                                               unsigned f, v2f &dp_, v2f &dq_) {
    const unsigned t = threadIdx.x;
    const int Lc = L - 1;
    const unsigned ti = (int)t < Lc ? t : 0u;
    const float2 *xf = x + (size_t)f * 4096;
    const float2 *prev = Lc ? (f ? xf - Lc : win + 1) : xf;
    const float2 *pa = prev + ti, *pb = xf + (4096 - (Lc ? Lc : 1) + ti);
    asm volatile("global_load_dwordx2 %0, %2, off\n\tglobal_load_dwordx2 %1, %3, off"
                 : "=&v"(dp_), "=&v"(dq_) : "v"(pa), "v"(pb) : "memory");
}
template <int INFLIGHT>
__device__ __forceinline__ void freq_wait_corr(v2f &dp_, v2f &dq_) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(dp_), "+v"(dq_) : "i"(INFLIGHT) : "memory");
}

// c_f[0 .. 256) of one frame into dst[t]: d, the balanced triangle sum, the fixed-order combination of the partial
// sums.  `wk` = 768 + 6*256 float2 of LDS work space (the exchange buffer), `gl` = the scaled taps g[0 .. 288) in
// LDS.  Three barriers, the last one after the partial sums have been read (`wk` is free on return).
__device__ __forceinline__ void freq_correction(float2 *__restrict__ wk, const float *__restrict__ gl, int L,
                                                v2f dp_, v2f dq_, float2 *__restrict__ dst) {
    const unsigned t = threadIdx.x;
    const int Lc = L - 1;
    {
        const v2f dd = dp_ - dq_;
        wk[pidx4(t)] = (int)t < Lc ? f2(dd) : make_float2(0.f, 0.f);
    }
    if (t < 64) wk[pidx4(256 + t)] = make_float2(0.f, 0.f);      // items reach sample 4q + 40c + 47 < 304
    __syncthreads();
    {
        // the lane's work item
        const unsigned ch = (t >= 64) + (t >= 118) + (t >= 162) + (t >= 196) + (t >= 220) + (t >= 234);
        if (t < 238) {
            const unsigned q = t - tri_base(ch), j0 = kTriS * ch;
            const unsigned m0 = 4 * q + j0;
            float2 acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = make_float2(0.f, 0.f);
            float2 D[8];
            {
                const float4 *dp = reinterpret_cast<const float4 *>(wk + pidx4(m0));
                const float4 q0 = dp[0], q1 = dp[1];
                D[0] = make_float2(q0.x, q0.y); D[1] = make_float2(q0.z, q0.w);
                D[2] = make_float2(q1.x, q1.y); D[3] = make_float2(q1.z, q1.w);
            }
#pragma unroll 5
            for (int jj = 0; jj < kTriS; jj += 4) {
                const float4 *dp = reinterpret_cast<const float4 *>(wk + pidx4(m0 + jj + 4));
                const float4 q0 = dp[0], q1 = dp[1];
                const float4 g4 = *reinterpret_cast<const float4 *>(gl + j0 + jj);
                D[4] = make_float2(q0.x, q0.y); D[5] = make_float2(q0.z, q0.w);
                D[6] = make_float2(q1.x, q1.y); D[7] = make_float2(q1.z, q1.w);
                const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[r].x = fmaf(gs[s], D[s + r].x, acc[r].x);
                        acc[r].y = fmaf(gs[s], D[s + r].y, acc[r].y);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) D[r] = D[4 + r];
            }
            float4 *cp = reinterpret_cast<float4 *>(wk + kOffPart + 6u * t);
            cp[0] = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
            cp[1] = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
        }
    }
    __syncthreads();
    {
        // output n = t: group t >> 2, partials of chunks 0 .. cnt-1, added in chunk order
        const unsigned rq = t >> 2, rcnt = (64u - rq + 9u) / 10u;
        const float2 *pp = wk + kOffPart + (t & 3u);
        float2 s = pp[6u * rq];                               // chunk 0: item rq
#pragma unroll
        for (unsigned c = 1; c < 7; ++c) {
            const float2 p = pp[6u * (tri_base(c) + (c < rcnt ? rq : 0u))];
            if (c < rcnt) s = make_float2(s.x + p.x, s.y + p.y);
        }
        dst[t] = s;
    }
    __syncthreads();
}

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

// hs = scale * FFT_4096{[h; 0]}; gcorr[j] = h[L-1-j] for j < L-1, zero beyond (256 floats); tw = the stream twiddle
// table (capi.hip make_stream_twiddles: W_4096^m, then rows W^{t 2^k} (k < 4), W^{(t&15)(t>>4)}, W_256^{(t&15)(t>>4)});
// win_next receives the L-sample filter window after the call (last L samples of x); null = the caller produces it.
__global__ void __launch_bounds__(256, 4)
firfft_crcf_4096_freq_kernel(const float2 *__restrict__ win, const float2 *__restrict__ x,
                             const float2 *__restrict__ hs, const float *__restrict__ gcorr, float scale, int L,
                             const float2 *__restrict__ tw, float2 *__restrict__ out,
                             float2 *__restrict__ win_next, unsigned nframes) {
    __shared__ __attribute__((aligned(16))) float2 lds[kFreqLdsFloat2];
    float2 *cvs = lds + kOffCvs;
    const float2 *T = lds + kOffT256;
    float *gl = reinterpret_cast<float *>(lds + kOffG);
    const float2 *ax = tw + 4096;
    const unsigned t = threadIdx.x, f = blockIdx.x;
    // ---- head: every load the frame needs up front, LDS written only after the last one has been issued ----
    const float g_t = gcorr[t];
    const float2 tw_t = ax[1280 + t];
    float2 v[16];
    v2f dp_, dq_;
    __builtin_amdgcn_sched_barrier(0);
    freq_load_corr(win, x, L, f, dp_, dq_);
    {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (size_t)f * 4096, 32768u);
#pragma unroll
        for (unsigned a = 0; a < 16; ++a) v[a] = buf_ld(rx, 8u * t, 2048u * a);     // in flight under the correction sum
    }
    __builtin_amdgcn_sched_barrier(0);
    gl[t] = g_t * scale;
    if (t < 32) gl[256 + t] = 0.f;
    lds[kOffT256 + (t >> 4) * kTRow + (t & 15u)] = tw_t;
    freq_wait_corr<16>(dp_, dq_);
    if (f == nframes - 1 && win_next) {              // new filter window = last L samples of the call (pipelined calls: written ahead by a launch of its own)
        const float2 *xf = x + (size_t)f * 4096;
        for (int i = t; i < L; i += 256) win_next[i] = xf[4096 - L + i];
    }
    // ---- correction c_f -> cvs (its first barrier also publishes gl and T) ----
    freq_correction(lds, gl, L, dp_, dq_, cvs);
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
        for (unsigned d = 0; d < 16; ++d) hv[d] = buf_ld(rh, 8u * t, 2048u * d);
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) v[d] = cmul(v[d], hv[d]);
    }
    // ---- correction transform, sum, store ----
    float2 u[16];
    freq_head256(cvs, u, lds, T, ax[1024 + t]);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)f * 4096, 32768u);
#pragma unroll
    for (unsigned d = 0; d < 16; ++d) buf_st(ro, 8u * t, 2048u * d, make_float2(v[d].x + u[d].x, v[d].y + u[d].y));
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

int launch_firfft_crcf_4096_freq(const cf32 *win, const cf32 *x, const cf32 *hs_scaled, const float *gcorr,
                                 float scale, int L, const cf32 *tw_stream, cf32 *spectra, cf32 *win_next,
                                 size_t nframes, hipStream_t st) {
    if (nframes == 0) return YAGI_OK;
    if (L < 1 || L > kFreqMaxTaps)
        return fail(YAGI_ERR_CONFIG, "frequency-domain stream kernel needs 1..%d taps (got %d)", kFreqMaxTaps, L);
    if (nframes > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    firfft_crcf_4096_freq_kernel<<<(unsigned)nframes, 256, 0, st>>>(
        reinterpret_cast<const float2 *>(win), reinterpret_cast<const float2 *>(x),
        reinterpret_cast<const float2 *>(hs_scaled), gcorr, scale, L, reinterpret_cast<const float2 *>(tw_stream),
        reinterpret_cast<float2 *>(spectra), reinterpret_cast<float2 *>(win_next), (unsigned)nframes);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

}  // namespace yagi
