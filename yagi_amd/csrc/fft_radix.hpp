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

// Odd-prime butterflies, natural order in and out.  X[k] and X[R-k] share their cosine part m_k and differ in the
// sign of the sine part n_k:  X[k] = m_k -+ i n_k, X[R-k] = m_k +- i n_k  (upper sign: forward).
template <int SIGN>
__device__ __forceinline__ void dft3(float2 (&v)[3]) {
    const float s1 = 0.86602540378443865f;                                  // sin(2 pi / 3)
    const float2 t1 = cadd(v[1], v[2]), u1 = csub(v[1], v[2]);
    const float2 m1 = f2(tov(v[0]) - 0.5f * tov(t1));
    v[0] = cadd(v[0], t1);
    addsub_rot<SIGN>(m1, cscale(u1, s1), v[1], v[2]);
}
template <int SIGN>
__device__ __forceinline__ void dft5(float2 (&v)[5]) {
    const float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;     // cos(2 pi k / 5)
    const float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;      // sin(2 pi k / 5)
    const v2f a = tov(v[0]);
    const v2f t1 = tov(v[1]) + tov(v[4]), t2 = tov(v[2]) + tov(v[3]);
    const v2f u1 = tov(v[1]) - tov(v[4]), u2 = tov(v[2]) - tov(v[3]);
    const v2f m1 = a + c1 * t1 + c2 * t2, m2 = a + c2 * t1 + c1 * t2;
    const v2f n1 = s1 * u1 + s2 * u2, n2 = s2 * u1 - s1 * u2;
    v[0] = f2(a + t1 + t2);
    addsub_rot<SIGN>(f2(m1), f2(n1), v[1], v[4]);
    addsub_rot<SIGN>(f2(m2), f2(n2), v[2], v[3]);
}
template <int SIGN>
__device__ __forceinline__ void dft7(float2 (&v)[7]) {
    const float c1 = 0.62348980185873353f, c2 = -0.22252093395631440f, c3 = -0.90096886790241913f;   // cos(2 pi k / 7)
    const float s1 = 0.78183148246802981f, s2 = 0.97492791218182361f, s3 = 0.43388373911755812f;    // sin(2 pi k / 7)
    const v2f a = tov(v[0]);
    const v2f t1 = tov(v[1]) + tov(v[6]), t2 = tov(v[2]) + tov(v[5]), t3 = tov(v[3]) + tov(v[4]);
    const v2f u1 = tov(v[1]) - tov(v[6]), u2 = tov(v[2]) - tov(v[5]), u3 = tov(v[3]) - tov(v[4]);
    const v2f m1 = a + c1 * t1 + c2 * t2 + c3 * t3;
    const v2f m2 = a + c2 * t1 + c3 * t2 + c1 * t3;
    const v2f m3 = a + c3 * t1 + c1 * t2 + c2 * t3;
    const v2f n1 = s1 * u1 + s2 * u2 + s3 * u3;
    const v2f n2 = s2 * u1 - s3 * u2 - s1 * u3;
    const v2f n3 = s3 * u1 - s1 * u2 + s2 * u3;
    v[0] = f2(a + t1 + t2 + t3);
    addsub_rot<SIGN>(f2(m1), f2(n1), v[1], v[6]);
    addsub_rot<SIGN>(f2(m2), f2(n2), v[2], v[5]);
    addsub_rot<SIGN>(f2(m3), f2(n3), v[3], v[4]);
}

template <int R, int SIGN>
__device__ __forceinline__ void dftR(float2 (&v)[R]) {
    if constexpr (R == 2) dft2<SIGN>(v[0], v[1]);
    else if constexpr (R == 3) dft3<SIGN>(v);
    else if constexpr (R == 5) dft5<SIGN>(v);
    else if constexpr (R == 7) dft7<SIGN>(v);
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

// The LAST pass (Ns R = N) with its outputs handed to `out(fr, k, value)` from registers instead of going back to LDS:
// lanes butterfly-fastest, so for every q the T = N/R lanes of a transform deliver T consecutive bins k = j + q T
// (a 128-byte run of a [frame][channel] row at T = 16).  Saves the pass's LDS writes, the barrier behind them and the
// read-back of the store loop.  Reads: s[r T] with consecutive j -> conflict-free within a transform.
template <int R, int SIGN, class Out>
__device__ __forceinline__ void stockham_last_pass_out(const float2 *__restrict__ src, int N, int nfr,
                                                       const float2 *__restrict__ twl, int tw_scale,
                                                       bool tw_is_forward, int pitch, Out out) {
    const int T = N / R;                                   // = Ns of the last pass: k = j
    const int lgT = 31 - __builtin_clz((unsigned)T);
    const int total = T * nfr;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int fr = e >> lgT, j = e & (T - 1);
        const float2 *s = src + fr * pitch + j;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = s[r * T];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            float2 w = twl[j * r * tw_scale];
            if (SIGN > 0 && tw_is_forward) w.y = -w.y;
            v[r] = cmul(v[r], w);
        }
        dftR<R, SIGN>(v);
#pragma unroll
        for (int q = 0; q < R; ++q) out(fr, j + q * T, v[dftR_pos<R>(q)]);
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

// Mixed-radix form of stockham_pass: N, Ns, T = N/R arbitrary (index math by division), table twiddles
// W_N^m with the direction's sign already applied.  R in {2,3,4,5,7,8,16}: register butterfly.
// tw_scale / conj_tw: the table is W_{N tw_scale}^m with forward sign and SIGN > 0 wants its conjugate (channelizers)
template <int R, int SIGN>
__device__ __forceinline__ void stockham_pass_any(const float2 *__restrict__ src, float2 *__restrict__ dst,
                                                  int N, int Ns, int nfr, const float2 *__restrict__ twl,
                                                  int tw_scale = 1, bool conj_tw = false) {
    const int T = N / R;
    const int total = T * nfr;
    const int tw_k = (N / (Ns * R)) * tw_scale;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int fr = e / T, j = e - fr * T;
        const int k = j % Ns;
        const float2 *s = src + fr * N + j;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = s[r * T];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) {
                float2 w = twl[k * r * tw_k];
                if (conj_tw) w.y = -w.y;
                v[r] = cmul(v[r], w);
            }
        }
        dftR<R, SIGN>(v);
        float2 *d = dst + fr * N + (j - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) d[q * Ns] = v[dftR_pos<R>(q)];
    }
}

// Any other radix (primes >= 11): every lane produces ONE output of a butterfly by a direct R-term sum with
// exact table twiddles (index arithmetic mod N): O(N R) work for this pass.
__device__ __forceinline__ void stockham_pass_direct(const float2 *__restrict__ src, float2 *__restrict__ dst,
                                                     int N, int R, int Ns, int nfr,
                                                     const float2 *__restrict__ twl) {
    const int T = N / R;
    const int tw_k = N / (Ns * R);                 // W_{Ns*R} = W_N^{tw_k}
    const int total = N * nfr;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int fr = e / N, idx = e - fr * N;
        const int q = idx / T, j = idx - q * T;
        const int k = j % Ns;
        // step = (k*tw_k + q*T) mod N, kept < N; the twiddle index advances by `step` per term
        const int step = (int)(((long long)k * tw_k + (long long)q * T) % N);
        const float2 *s = src + fr * N;
        int m = 0;
        float2 acc = s[j];
        for (int r = 1; r < R; ++r) {
            m += step;
            if (m >= N) m -= N;
            acc = cadd(acc, cmul(s[j + r * T], twl[m]));
        }
        dst[fr * N + (j / Ns) * Ns * R + k + q * Ns] = acc;
    }
}

// LDS pitch (float2) of one transform slot when nfr = 2^k <= 32 transforms of <= N points share a pass (FRFAST)
__host__ __device__ inline int frfast_pitch(int N, int nfr) { return N + 32 / nfr; }

// ---------------------------------------------------------------------------------------------
// N = 256 M (M in {1, 2, 4, 8}) in registers: the 4096-point scheme with a radix-M last pass.  A 256-lane
// workgroup owns 16/M transforms; transform tr = t / 16M, lane u = t % 16M within it.
//   in : v[a] = x[16M a + u], a < 16
//   out: v[i M + d] = X[u + 16M i + 256 d],  i < 16/M, d < M
// `lds` = kFft4096LdsFloat2 float2.  Exchange strides 17M and 256 + 16/M keep all LDS accesses bank-conflict free.
// M <= 4: a transform's 16M lanes are one wave (or a part of one) and both exchanges stay inside the transform's own
// 272M-float2 slot, so the exchanges order themselves with wave barriers (the LDS serves a wave's accesses in issue
// order) and the function holds no __syncthreads(); at M = 1 the second exchange would hand every lane its own values
// back (pass 2 already leaves X[u + 16 c'] in lane u) and is a register renaming instead.  M = 8: a transform spans two
// waves, 3 __syncthreads().  Every lane must call it; the caller puts a __syncthreads() between this function and its
// own use of `lds` either side.
template <int SIGN, int M>
__device__ __forceinline__ void fft_n256m_passes_to_regs(float2 (&v)[16], float2 *__restrict__ lds,
                                                         const float2 *__restrict__ tw) {
    constexpr int N = 256 * M, LT = 16 * M, B = 16 / M;
    constexpr int S1 = 17 * M, T1 = 16 * S1;                    // exchange 1: row stride, transform stride
    constexpr int S2 = 256 + 16 / M, T2 = M * S2;               // exchange 2
    constexpr bool kInWave = LT <= 64;
    constexpr int T2s = kInWave ? T1 : T2;                      // in-wave: exchange 2 in the transform's exchange-1 slot
    static_assert(T2 <= T1 && B * T1 <= kFft4096LdsFloat2 && B * T2 <= kFft4096LdsFloat2, "LDS layout");
    auto sync = [] {
        if constexpr (kInWave) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    };
    const unsigned t = threadIdx.x, tr = t / LT, u = t % LT;
    // ---- pass 1 ----
    dft16<SIGN>(v);
    {
        float2 w[16];
        twiddle_powers(w, tw, u, (unsigned)(N - 1));             // W_N^{u c}: u c < N, the mask never wraps
        float2 *e1 = lds + tr * T1 + u;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float2 z = v[dft16_pos(c)];
            if (c) z = cmul(z, w[c]);
            e1[c * S1] = z;
        }
    }
    sync();
    // ---- pass 2 ----
    {
        const unsigned c = u / M, bp = u % M;
        const float2 *e1 = lds + tr * T1 + c * S1 + bp;
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = e1[M * a];
        sync();                                                  // exchange-1 reads done before the buffer is reused
        dft16<SIGN>(v);
        if constexpr (M == 1) {                                  // b' = 0: every twiddle of this pass is 1, and lane u = c
            float2 r[16];                                        // already holds X[u + 16 c']
#pragma unroll
            for (int cp = 0; cp < 16; ++cp) r[cp] = v[dft16_pos(cp)];
#pragma unroll
            for (int cp = 0; cp < 16; ++cp) v[cp] = r[cp];
            return;
        } else {
            float2 *e2 = lds + tr * T2s + bp * S2 + c;
            float2 w[16];
            twiddle_powers(w, tw, 16 * bp, (unsigned)(N - 1));  // W_{16M}^{b' c'} = W_N^{16 b' c'}
#pragma unroll
            for (int cp = 0; cp < 16; ++cp) {
                float2 z = v[dft16_pos(cp)];
                if (cp) z = cmul(z, w[cp]);
                e2[16 * cp] = z;
            }
        }
    }
    sync();
    // ---- pass 3: 16/M radix-M butterflies per lane ----
    {
        const float2 *e2 = lds + tr * T2s + u;
#pragma unroll
        for (int i = 0; i < B; ++i) {
            float2 r[M];
#pragma unroll
            for (int bp = 0; bp < M; ++bp) r[bp] = e2[bp * S2 + LT * i];
            if constexpr (M > 1) dftR<M, SIGN>(r);
#pragma unroll
            for (int d = 0; d < M; ++d) v[i * M + d] = r[dftR_pos<M>(d)];
        }
    }
    sync();                                                      // in-wave: the caller's next writes stay behind these reads
}

}  // namespace yagi
