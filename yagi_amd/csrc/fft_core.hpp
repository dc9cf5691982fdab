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

template <int SIGN>
__device__ __forceinline__ void dft4(float2 &a, float2 &b, float2 &c, float2 &d) {
    const float2 s0 = cadd(a, c), d0 = csub(a, c), s1 = cadd(b, d), d1 = csub(b, d);
    const float2 r = (SIGN < 0) ? mul_mi(d1) : mul_pi(d1);
    a = cadd(s0, s1);
    c = csub(s0, s1);
    b = cadd(d0, r);
    d = csub(d0, r);
}

template <int SIGN>
__device__ __forceinline__ float2 w16(float c, float s) { return make_float2(c, SIGN < 0 ? -s : s); }

// In-place 16-point DFT.  Input v[n]; output X[k] is left at v[4*(k&3) + (k>>2)].
template <int SIGN>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4<SIGN>(v[n0], v[4 + n0], v[8 + n0], v[12 + n0]);
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r2 = 0.70710678118654752f;
    // v[4*k0 + n0] *= W16^{n0*k0}
    v[4 * 1 + 1] = cmul(v[4 * 1 + 1], w16<SIGN>(c1, s1));        // m = 1
    v[4 * 2 + 1] = cmul(v[4 * 2 + 1], w16<SIGN>(r2, r2));        // m = 2
    v[4 * 3 + 1] = cmul(v[4 * 3 + 1], w16<SIGN>(s1, c1));        // m = 3
    v[4 * 1 + 2] = cmul(v[4 * 1 + 2], w16<SIGN>(r2, r2));        // m = 2
    v[4 * 2 + 2] = (SIGN < 0) ? mul_mi(v[4 * 2 + 2]) : mul_pi(v[4 * 2 + 2]);   // m = 4
    v[4 * 3 + 2] = cmul(v[4 * 3 + 2], w16<SIGN>(-r2, r2));       // m = 6
    v[4 * 1 + 3] = cmul(v[4 * 1 + 3], w16<SIGN>(s1, c1));        // m = 3
    v[4 * 2 + 3] = cmul(v[4 * 2 + 3], w16<SIGN>(-r2, r2));       // m = 6
    v[4 * 3 + 3] = cmul(v[4 * 3 + 3], w16<SIGN>(-c1, -s1));      // m = 9
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) dft4<SIGN>(v[4 * k0], v[4 * k0 + 1], v[4 * k0 + 2], v[4 * k0 + 3]);
}

// position of output k inside v after dft16
__device__ __forceinline__ constexpr int dft16_pos(int k) { return 4 * (k & 3) + (k >> 2); }

// Passes 1..3 of the 4096-point transform.  On entry lane b (= threadIdx.x, 0..255) holds
// v[a] = x[256a + b].  `lds` is a kFft4096LdsFloat2 float2 buffer nobody else touches; `tw` is
// the W_4096^m table (sign already per direction).  Output goes to out[k], k in [0,4096).
// Contains 4 __syncthreads(); all 256 lanes must call it.
template <int SIGN, bool GUARD = false>
__device__ __forceinline__ void fft4096_passes(float2 (&v)[16], float2 *__restrict__ lds,
                                               const float2 *__restrict__ tw,
                                               float2 *__restrict__ out) {
    // GUARD lets a workgroup larger than 256 lanes call this: lanes >= 256 only join the barriers.
    const unsigned t = threadIdx.x;
    const bool active = !GUARD || t < 256;
    // ---- pass 1 (lane b = t) ----
    if (active) {
        dft16<SIGN>(v);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float2 z = v[dft16_pos(c)];
            if (c) z = cmul(z, tw[(unsigned)(t * c)]);
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
#pragma unroll
            for (int cp = 0; cp < 16; ++cp) {
                float2 u = v[dft16_pos(cp)];
                if (cp) u = cmul(u, tw[(unsigned)(16 * bp * cp)]);
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
        for (int d = 0; d < 16; ++d) out[t + 256 * d] = v[dft16_pos(d)];
    }
    __syncthreads();                           // LDS free for the caller's next transform
}

}  // namespace yagi
