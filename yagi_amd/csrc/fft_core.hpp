// fft_core.hpp -- device building blocks of the 4096-point complex FFT (Fft::run semantics,
// src/fft/mod.rs:19-26,45-48: unnormalised, forward = e^{-j 2 pi n k / N}); shared by the batch
// kernel (fft_kernels.hip) and the fused FIR->FFT stream kernel (stream_kernels.hip).
//
// One 256-lane workgroup owns one transform.  4096 = 16 x 16 x 16: three passes of 16-point
// register butterflies (two radix-4 stages each), two exchanges through one 34 KiB LDS buffer.
//   pass 1  lane b        : v[a]  = x[256a + b]            -> Z_c[b]  = W4096^{bc} sum_a v[a] W16^{ac}
//   pass 2  lane (c,b')   : v[a'] = Z_c[16a' + b']         -> U_{c,c'}[b'] = W256^{b'c'} sum_a' ...
//   pass 3  lane c + 16c' : v[b'] = U_{c,c'}[b']           -> X[c + 16c' + 256d'] = sum_b' v[b'] W16^{b'd'}
// Global reads in pass 1 and writes in pass 3 are both 512 B contiguous per wave instruction.
// LDS layouts (float2 units), chosen so every ds access below is bank-conflict free:
//   exchange 1: [c][b] with row stride 272   (pass-2 reads: 16 lanes b' contiguous, c rows 32 banks apart)
//   exchange 2: [b'][c'*16+c] with row stride 257 (pass-2 writes: 16 lanes b' -> distinct even banks)
#pragma once
#include "devmath.hpp"

namespace yagi {

constexpr int kFft4096LdsFloat2 = 16 * 272;          // 34816 B
constexpr int kEx1Stride = 272;
constexpr int kEx2Stride = 257;

// ROTC: input c is first multiplied by -+i (a W^{N/4} twiddle folded into the butterfly's first adds)
template <int SIGN, bool ROTC = false>
__device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d) {
    float2 s0, d0;
    if (ROTC) addsub_rot<SIGN>(a, c, s0, d0);
    else { s0 = cadd(a, c); d0 = csub(a, c); }
    const float2 s1 = cadd(b, d), d1 = csub(b, d);
    a = cadd(s0, s1);
    c = csub(s0, s1);
    addsub_rot<SIGN>(d0, d1, b, d);
}

template <int SIGN>
__device__ __forceinline__ float2 w16(float c, float s) { return make_float2(c, SIGN < 0 ? -s : s); }

// In-place 16-point DFT.  Input v[n]; output X[k] is left at v[4*(k&3) + (k>>2)].
// 64 packed adds + 8 constant products (2 packed instructions each) = 80 VALU instructions.
template <int SIGN>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4<SIGN>(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r2 = 0.70710678118654752f;
    // v[4*k0 + n0] *= W16^{n0*k0}
    v[4 * 1 + 1] = cmul_k(v[4 * 1 + 1], w16<SIGN>(c1, s1));        // m = 1
    v[4 * 2 + 1] = cmul_k(v[4 * 2 + 1], w16<SIGN>(r2, r2));        // m = 2
    v[4 * 3 + 1] = cmul_k(v[4 * 3 + 1], w16<SIGN>(s1, c1));        // m = 3
    v[4 * 1 + 2] = cmul_k(v[4 * 1 + 2], w16<SIGN>(r2, r2));        // m = 2
    //  v[4 * 2 + 2] *= W16^4 = -+i: inside the k0 = 2 butterfly below (ROTC)
    v[4 * 3 + 2] = cmul_k(v[4 * 3 + 2], w16<SIGN>(-r2, r2));       // m = 6
    v[4 * 1 + 3] = cmul_k(v[4 * 1 + 3], w16<SIGN>(s1, c1));        // m = 3
    v[4 * 2 + 3] = cmul_k(v[4 * 2 + 3], w16<SIGN>(-r2, r2));       // m = 6
    v[4 * 3 + 3] = cmul_k(v[4 * 3 + 3], w16<SIGN>(-c1, -s1));      // m = 9
    dft4<SIGN>(v[0], v[1], v[2], v[3]);
    dft4<SIGN>(v[4], v[5], v[6], v[7]);
    dft4<SIGN, true>(v[8], v[9], v[10], v[11]);
    dft4<SIGN>(v[12], v[13], v[14], v[15]);
}

// position of output k inside v after dft16
__device__ __forceinline__ constexpr int dft16_pos(int k) { return 4 * (k & 3) + (k >> 2); }

// Twiddles w[c] = W^(c*m), c = 1..15, from FOUR exact table entries (c = 1, 2, 4, 8) and products of
// at most three of them (depth <= 2 multiplications for 13 of the 15, 3 for w[15]).  A lane needs 15
// twiddles per pass; fetching each from the table is a 64-address gather per wave instruction and made
// the L1/TA path -- not HBM, LDS or FP32 -- the bottleneck of the transform.
__device__ __forceinline__ void twiddle_powers(float2 (&w)[16], const float2 *__restrict__ tw, unsigned m,
                                               unsigned mask) {
    w[1] = tw[m & mask];
    w[2] = tw[(2 * m) & mask];
    w[4] = tw[(4 * m) & mask];
    w[8] = tw[(8 * m) & mask];
    w[3] = cmul(w[2], w[1]);
    w[5] = cmul(w[4], w[1]);
    w[6] = cmul(w[4], w[2]);
    w[7] = cmul(w[4], w[3]);
    w[9] = cmul(w[8], w[1]);
    w[10] = cmul(w[8], w[2]);
    w[11] = cmul(w[8], w[3]);
    w[12] = cmul(w[8], w[4]);
    w[13] = cmul(w[8], w[5]);
    w[14] = cmul(w[8], w[6]);
    w[15] = cmul(w[8], w[7]);
}

// the same from four given exact powers W^m, W^2m, W^4m, W^8m
__device__ __forceinline__ void twiddle_powers_from(float2 (&w)[16], float2 w1, float2 w2, float2 w4, float2 w8) {
    w[1] = w1; w[2] = w2; w[4] = w4; w[8] = w8;
    w[3] = cmul(w[2], w[1]);
    w[5] = cmul(w[4], w[1]);
    w[6] = cmul(w[4], w[2]);
    w[7] = cmul(w[4], w[3]);
    w[9] = cmul(w[8], w[1]);
    w[10] = cmul(w[8], w[2]);
    w[11] = cmul(w[8], w[3]);
    w[12] = cmul(w[8], w[4]);
    w[13] = cmul(w[8], w[5]);
    w[14] = cmul(w[8], w[6]);
    w[15] = cmul(w[8], w[7]);
}

// Passes 1..3 of the 4096-point transform.  On entry lane b (= threadIdx.x, 0..255) holds
// v[a] = x[256a + b].  `lds` is a kFft4096LdsFloat2 float2 buffer nobody else touches; `tw` is
// the W_4096^m table (sign already per direction).  Output goes to out[k], k in [0,4096).
// Contains 4 __syncthreads(); all 256 lanes must call it.
// TWP = twiddles by powers (4 exact loads + products); TWP = false keeps the 15 table gathers per pass
// (an ablation knob: the gathers were faster only while the fused kernels sat at 2 waves/SIMD).
template <int SIGN, bool GUARD = false, bool TWP = true>
__device__ __forceinline__ void fft4096_passes(float2 (&v)[16], float2 *__restrict__ lds,
                                               const float2 *__restrict__ tw,
                                               float2 *__restrict__ out) {
    // GUARD lets a workgroup larger than 256 lanes call this: lanes >= 256 only join the barriers.
    const unsigned t = threadIdx.x;
    const bool active = !GUARD || t < 256;
    // ---- pass 1 (lane b = t) ----
    if (active) {
        dft16<SIGN>(v);
        float2 w[16];
        if (TWP) twiddle_powers(w, tw, t, 4095u);  // W4096^(t*c): t*c < 4096, the mask never wraps
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float2 z = v[dft16_pos(c)];
            if (c) z = cmul(z, TWP ? w[c] : tw[(unsigned)(t * c)]);
            lds[c * kEx1Stride + t] = z;
        }
    }
    __syncthreads();
    // ---- pass 2 (lane = c*16 + b') ----
    {
        const unsigned c = t >> 4, bp = t & 15;
        if (active) {
#pragma unroll
            for (int a = 0; a < 16; ++a) v[a] = lds[c * kEx1Stride + 16 * a + bp];
        }
        __syncthreads();                       // exchange-1 reads done before the buffer is reused
        if (active) {
            dft16<SIGN>(v);
            float2 w[16];
            if (TWP) twiddle_powers(w, tw, 16 * bp, 4095u);   // W256^(b'*c') = W4096^(16 b' c') < 3600
#pragma unroll
            for (int cp = 0; cp < 16; ++cp) {
                float2 u = v[dft16_pos(cp)];
                if (cp) u = cmul(u, TWP ? w[cp] : tw[(unsigned)(16 * bp * cp)]);
                lds[bp * kEx2Stride + cp * 16 + c] = u;
            }
        }
    }
    __syncthreads();
    // ---- pass 3 (lane = c + 16c') ----
    if (active) {
#pragma unroll
        for (int bp = 0; bp < 16; ++bp) v[bp] = lds[bp * kEx2Stride + t];
        dft16<SIGN>(v);
#pragma unroll
        for (int d = 0; d < 16; ++d) st_stream(out + t + 256 * d, v[dft16_pos(d)]);
    }
    __syncthreads();                           // LDS free for the caller's next transform
}

// Passes 2 and 3 with the result left in registers: exchange-1 data (pass-1 output, lds[c*272 + b])
// has been stored and a __syncthreads() has followed.  On return lane t holds X[t + 256 d] in v[d].
template <int SIGN, bool TWP = true>
__device__ __forceinline__ void fft4096_pass23_to_regs(float2 (&v)[16], float2 *__restrict__ lds,
                                                       const float2 *__restrict__ tw) {
    const unsigned t = threadIdx.x;
    {
        const unsigned c = t >> 4, bp = t & 15;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = lds[c * kEx1Stride + 16 * a + bp];
        __syncthreads();
        dft16<SIGN>(v);
        float2 wt[16];
        if (TWP) twiddle_powers(wt, tw, 16 * bp, 4095u);
#pragma unroll
        for (int cp = 0; cp < 16; ++cp) {
            float2 u = v[dft16_pos(cp)];
            if (cp) u = cmul(u, TWP ? wt[cp] : tw[(unsigned)(16 * bp * cp)]);
            lds[bp * kEx2Stride + cp * 16 + c] = u;
        }
    }
    __syncthreads();
    float2 w[16];
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) w[bp] = lds[bp * kEx2Stride + t];
    dft16<SIGN>(w);
#pragma unroll
    for (int d = 0; d < 16; ++d) v[d] = w[dft16_pos(d)];
    __syncthreads();
}

// Same three passes as fft4096_passes, but the result stays in registers: on return lane t holds
// X[t + 256 d] in v[d] (d = 0..15) -- which is exactly the layout pass 1 expects, so transforms can be
// chained (FFT -> pointwise product -> inverse FFT) without touching LDS or HBM in between.
template <int SIGN, bool TWP = true>
__device__ __forceinline__ void fft4096_passes_to_regs(float2 (&v)[16], float2 *__restrict__ lds,
                                                       const float2 *__restrict__ tw) {
    const unsigned t = threadIdx.x;
    dft16<SIGN>(v);
    {
        float2 w[16];
        if (TWP) twiddle_powers(w, tw, t, 4095u);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float2 z = v[dft16_pos(c)];
            if (c) z = cmul(z, TWP ? w[c] : tw[(unsigned)(t * c)]);
            lds[c * kEx1Stride + t] = z;
        }
    }
    __syncthreads();
    fft4096_pass23_to_regs<SIGN, TWP>(v, lds, tw);
}

// ---------------------------------------------------------------------------------------------
// The same three passes with the twiddle scheme of the headline kernel (freq_kernels.hip), either direction:
//   pass 1  W_4096^{tc}: four exact entries W^{t 2^k} from lane-indexed table rows (coalesced loads) + products;
//   pass 2  W_256^{b'c'} = T[b'][c'] from a 16 x 16 table in LDS (row pitch 17: immediate-offset ds_reads, no gathers).
// `ax` = the auxiliary rows behind the W_4096 table of make_stream_twiddles (capi.hip): rows 0-3 W^{t 2^k}, row 5
// W_256^{(t&15)(t>>4)} (the T table's source), FORWARD values -- the inverse direction multiplies by their conjugates.
// fft4096_tab_load fills T (call once per workgroup; a barrier must follow before T is read).
// ---------------------------------------------------------------------------------------------
constexpr unsigned kTRow = 17;
constexpr unsigned kFft4096TabFloat2 = 16 * kTRow;
__device__ __forceinline__ void fft4096_tab_load(float2 *__restrict__ T, const float2 *__restrict__ ax) {
    const unsigned t = threadIdx.x;
    T[(t >> 4) * kTRow + (t & 15u)] = ax[1280 + t];
}
template <int SIGN>
__device__ __forceinline__ void fft4096_tab_to_regs(float2 (&v)[16], float2 *__restrict__ lds, const float2 *__restrict__ T,
                                                    const float2 *__restrict__ ax) {
    const unsigned t = threadIdx.x;
    dft16<SIGN>(v);
    {
        float2 w[16];
        twiddle_powers_from(w, ax[t], ax[256 + t], ax[512 + t], ax[768 + t]);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float2 z = v[dft16_pos(c)];
            if (c) z = cmul_dir<SIGN>(z, w[c]);
            lds[c * kEx1Stride + t] = z;
        }
    }
    __syncthreads();
    {
        const unsigned c = t >> 4, bp = t & 15;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = lds[c * kEx1Stride + 16 * a + bp];
        __syncthreads();
        dft16<SIGN>(v);
#pragma unroll
        for (int cp = 0; cp < 16; ++cp) {
            float2 u = v[dft16_pos(cp)];
            if (cp) u = cmul_dir<SIGN>(u, T[bp * kTRow + cp]);
            lds[bp * kEx2Stride + cp * 16 + c] = u;
        }
    }
    __syncthreads();
    float2 w[16];
#pragma unroll
    for (int bp = 0; bp < 16; ++bp) w[bp] = lds[bp * kEx2Stride + t];
    dft16<SIGN>(w);
#pragma unroll
    for (int d = 0; d < 16; ++d) v[d] = w[dft16_pos(d)];
    __syncthreads();
}

}  // namespace yagi
