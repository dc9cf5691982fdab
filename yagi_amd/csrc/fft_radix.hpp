// fft_radix.hpp -- register butterflies (radix 2/4/8/16) and one Stockham autosort pass over
// transforms held in LDS.  Used for every power-of-two size other than the dedicated 4096 kernel:
// Fft::run for N = 2^k <= 8192 (fft_kernels.hip) and the M-point DFTs inside the polyphase
// channelizers (chan_kernels.hip).  Definition as in src/fft/mod.rs:19-26 (unnormalised, forward =
// e^{-j 2 pi n k / N}; SIGN = -1 forward, +1 backward).
#pragma once
#include "fft_core.hpp"

namespace yagi {

template <int SIGN>
__device__ __forceinline__ void dft2(float2 &a, float2 &b) {
    const float2 t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// In-place 8-point DFT, natural order in and out.
template <int SIGN>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    // n = 4*n1 + n0 (n1 < 2, n0 < 4): 2-point DFTs over n1, twiddle W8^{n0*k1}, 4-point DFTs over n0
    const float r2 = 0.70710678118654752f;
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft2<SIGN>(v[n0], v[4 + n0]);      // v[4*k1 + n0]
    v[5] = cmul_k(v[5], w16<SIGN>(r2, r2));                             // W8^1
    v[6] = (SIGN < 0) ? mul_mi(v[6]) : mul_pi(v[6]);                  // W8^2
    v[7] = cmul_k(v[7], w16<SIGN>(-r2, r2));                            // W8^3
    dft4<SIGN>(v[0], v[1], v[2], v[3]);                               // -> X[k1=0 + 2*k0] at v[k0]
    dft4<SIGN>(v[4], v[5], v[6], v[7]);                               // -> X[1 + 2*k0] at v[4 + k0]
    // reorder to natural: X[k] = v[4*(k&1) + (k>>1)]
    const float2 x1 = v[4], x2 = v[1], x3 = v[5], x4 = v[2], x5 = v[6], x6 = v[3];
    v[1] = x1; v[2] = x2; v[3] = x3; v[4] = x4; v[5] = x5; v[6] = x6;
}

template <int R, int SIGN>
__device__ __forceinline__ void dftR(float2 (&v)[R]) {
    if constexpr (R == 2) dft2<SIGN>(v[0], v[1]);
    else if constexpr (R == 4) dft4<SIGN>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) dft8<SIGN>(v);
    else {
        static_assert(R == 16, "radix");
        dft16<SIGN>(v);
    }
}
template <int R>
__device__ __forceinline__ constexpr int dftR_pos(int k) { return R == 16 ? dft16_pos(k) : k; }

// One Stockham pass of radix R over `nfr` independent N-point transforms stored [frame][pitch] in LDS.
//   butterfly j in [0, N/R), k = j mod Ns:
//     dst[(j/Ns)*Ns*R + k + q*Ns] = sum_r src[j + r*N/R] * W_{Ns*R}^{k*r} * W_R^{r*q}
// twl = LDS (or global) table of W_Ntab^m, N * tw_scale == Ntab.  Ns, N powers of two.
// Lane -> (transform, butterfly) mapping:
//   FRFAST = false: butterfly index fastest (pitch = N).  Fine while one transform's butterflies fill a lane
//                   group; with many short transforms the lanes of a group sit N float2 apart = on the same
//                   banks (8-way conflicts for 64-point transforms: SQ_LDS_BANK_CONFLICT was 80 % of the
//                   channelizers' LDS cycles).
//   FRFAST = true:  transform index fastest, nfr = 2^lgnfr <= 32, pitch = N + 32/nfr: the 32 lanes of a group
//                   are nfr transforms x 32/nfr consecutive butterflies, 2*pitch = 2*32/nfr (mod 64) words
//                   apart -> every read and every last-pass write is conflict-free.
template <int R, int SIGN, bool FRFAST = false>
__device__ __forceinline__ void stockham_pass(const float2 *__restrict__ src, float2 *__restrict__ dst,
                                              int N, int Ns, int nfr, const float2 *__restrict__ twl,
                                              int tw_scale, bool tw_is_forward = true, int pitch = 0,
                                              int lgnfr = 0) {
    const int T = N / R;
    const int lgT = 31 - __builtin_clz((unsigned)T);       // N, R powers of two: index math by shifts
    const int total = T * nfr;
    const int tw_k = (N / (Ns * R)) * tw_scale;
    if (!FRFAST) pitch = N;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int fr = FRFAST ? (e & (nfr - 1)) : (e >> lgT), j = FRFAST ? (e >> lgnfr) : (e & (T - 1));
        const int k = j & (Ns - 1);
        const float2 *s = src + fr * pitch + j;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = s[r * T];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                float2 w = twl[k * r * tw_k];
                if (SIGN > 0 && tw_is_forward) w.y = -w.y;    // forward-sign table used for a backward transform
                v[r] = cmul(v[r], w);
            }
        }
        dftR<R, SIGN>(v);
        float2 *d = dst + fr * pitch + (j - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) d[q * Ns] = v[dftR_pos<R>(q)];
    }
}

// radix plan for N = 2^lg: the fewest passes radix <= 16 allows (ceil(lg/4)), with the bits spread evenly over
// them (64 = 8 x 8 rather than 16 x 4: every pass then has N/8 butterflies per transform, so short transforms
// keep all lanes of the workgroup busy in every pass)
struct Pow2Plan { int n; int r[8]; };
inline Pow2Plan make_pow2_plan(int N) {
    Pow2Plan p{0, {0}};
    int lg = 0;
    while ((1 << lg) < N) ++lg;
    if (lg == 0) return p;
    const int np = (lg + 3) / 4;
    for (int i = 0; i < np; ++i) {
        const int bits = lg / np + (i < lg % np ? 1 : 0);
        p.r[p.n++] = 1 << bits;
    }
    return p;
}

// all passes of `plan` over nfr transforms; returns the buffer holding the result.
// Every lane of the workgroup must call it (one __syncthreads per pass).
template <int SIGN, bool FRFAST = false>
__device__ __forceinline__ float2 *lds_fft_pow2(float2 *src, float2 *dst, int N, int nfr,
                                                const Pow2Plan &plan, const float2 *__restrict__ twl,
                                                int tw_scale, bool tw_is_forward = true, int pitch = 0,
                                                int lgnfr = 0) {
    int Ns = 1;
    for (int f = 0; f < plan.n; ++f) {
        const int R = plan.r[f];
        if (R == 16) stockham_pass<16, SIGN, FRFAST>(src, dst, N, Ns, nfr, twl, tw_scale, tw_is_forward, pitch, lgnfr);
        else if (R == 8) stockham_pass<8, SIGN, FRFAST>(src, dst, N, Ns, nfr, twl, tw_scale, tw_is_forward, pitch, lgnfr);
        else if (R == 4) stockham_pass<4, SIGN, FRFAST>(src, dst, N, Ns, nfr, twl, tw_scale, tw_is_forward, pitch, lgnfr);
        else stockham_pass<2, SIGN, FRFAST>(src, dst, N, Ns, nfr, twl, tw_scale, tw_is_forward, pitch, lgnfr);
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
        Ns *= R;
    }
    return src;
}

// LDS pitch (float2) of one transform slot when nfr = 2^k <= 32 transforms of <= N points share a pass (FRFAST)
__host__ __device__ inline int frfast_pitch(int N, int nfr) { return N + 32 / nfr; }

}  // namespace yagi
