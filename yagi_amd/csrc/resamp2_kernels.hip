// resamp2_kernels.hip -- Resamp2<T,Coeff> block kernels (src/filter/resampler/resamp2.rs:104-180): the half-band
// filter / two-channel analysis / synthesis bank / decimator / interpolator, one launch per block.
//
// Every form of the reference is "a 2m-tap branch filter on one polyphase stream beside a pure delay on the other":
//     F(S, p) = sum_{k < 2m} h1[k] S[p - (2m-1) + k]        (h1.dotprod(window.read()): oldest sample first)
//     D(S, p) = S[p - m]                                    (window.index(m-1) after the push)
// with the streams S0 (window w0) and S1 (window w1) formed from the block's input:
//     decim        S1[u] = x[2u],        S0[u] = x[2u+1]        y[u]    = (D(S0,u) + F(S1,u)) scale
//     analyzer     S1[u] = x[2u]/2,      S0[u] = x[2u+1]/2      y[2u]   = (F(S1,u) + D(S0,u)) scale,  y[2u+1] = (F - D) scale
//     synthesizer  S0[u] = x[2u]+x[2u+1], S1[u] = x[2u]-x[2u+1] y[2u]   = D(S0,u) scale,              y[2u+1] = F(S1,u) scale
//     interp       S0[u] = S1[u] = x[u]                          y[2u]   = D(S0,u) scale,              y[2u+1] = F(S1,u) scale
//     filter       samples alternate between the windows (toggle): with A the window the block's first sample enters
//                  and B the other, A[u] = x[2u], B[u] = x[2u+1]:
//                  sample 2u:   yi = D(A,u), yq = F(B,u-1);   sample 2u+1: yi = D(B,u), yq = F(A,u)
//                  y0 = (yi + yq)/2 scale, y1 = (yi - yq)/2 scale, stored as (y0, y1) pairs
// Stream indices below zero read the object's state: the two windows (2m samples each, oldest first) kept in HBM.
// A workgroup stages 1024 units of both streams (+ 2m of history) in LDS and every lane walks the 2m taps (wave-uniform:
// scalar loads) for its four units, 256 apart; the windows after the block are written by a second, tiny launch.
#include "devmath.hpp"
#include "kernels.hpp"

namespace yagi {

constexpr int kR2Lanes = 256, kR2Upl = 4, kR2Tile = kR2Lanes * kR2Upl;   // lanes, units per lane, units per workgroup

__device__ __forceinline__ float r2_sub(float a, float b) { return a - b; }
__device__ __forceinline__ cf32 r2_sub(cf32 a, cf32 b) { return cf32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ float r2_half(float a) { return 0.5f * a; }
__device__ __forceinline__ cf32 r2_half(cf32 a) { return cf32{0.5f * a.re, 0.5f * a.im}; }

// value of stream s (0 / 1) at block-relative index u >= 0
template <class T, int MODE>
__device__ __forceinline__ T r2_stream(const T *__restrict__ x, size_t nx, int s, long long u, int c0) {
    if (MODE == kR2Interp) return (size_t)u < nx ? x[u] : zero_of<T>();
    if (MODE == kR2Filter) {
        // window s receives the even samples of the block iff s == c0 (c0 = toggle at the block's start)
        const size_t j = 2 * (size_t)u + (s == c0 ? 0 : 1);
        return j < nx ? x[j] : zero_of<T>();
    }
    const size_t j = 2 * (size_t)u;
    if (j + 1 >= nx) return zero_of<T>();               // beyond the block (staging of the last tile)
    const T a = x[j], b = x[j + 1];
    if (MODE == kR2Decim) return s ? a : b;
    if (MODE == kR2Analyzer) return r2_half(s ? a : b);
    return s ? r2_sub(a, b) : add(a, b);                 // synthesizer
}

template <class T, class C, int MODE>
__global__ void __launch_bounds__(kR2Lanes)
resamp2_kernel(const T *__restrict__ state, const T *__restrict__ x, size_t nx, const C *__restrict__ h1, int m,
               C scale, int c0, T *__restrict__ y, size_t nunits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char r2_lds[];
    const int m2 = 2 * m, span = kR2Tile + m2;
    T *S0 = reinterpret_cast<T *>(r2_lds), *S1 = S0 + span;
    const long long tile0 = (long long)blockIdx.x * kR2Tile;
    // stage both streams: LDS index j <-> stream index tile0 - 2m + j
    for (int j = threadIdx.x; j < span; j += kR2Lanes) {
        const long long u = tile0 - m2 + j;
        T a, b;
        if (u < 0) {
            a = state[m2 + u];
            b = state[m2 + m2 + u];
        } else {
            a = r2_stream<T, MODE>(x, nx, 0, u, c0);
            b = r2_stream<T, MODE>(x, nx, 1, u, c0);
        }
        S0[j] = a;
        S1[j] = b;
    }
    __syncthreads();
    // F(S, q) with q given as an LDS index: taps oldest first
    auto fir = [&](const T *S, int q) {
        T acc = zero_of<T>();
        const T *w = S + (q - (m2 - 1));
        for (int k = 0; k < m2; ++k) acc = mac(acc, w[k], h1[k]);
        return acc;
    };
#pragma unroll
    for (int uq = 0; uq < kR2Upl; ++uq) {
        const int lu = (int)threadIdx.x + kR2Lanes * uq;     // unit within the tile
        const size_t i = (size_t)tile0 + lu;
        if (i >= nunits) break;
        const int p = m2 + lu;                               // LDS index of stream index i
        if (MODE == kR2Decim) {
            y[i] = mul(add(S0[p - m], fir(S1, p)), scale);
        } else if (MODE == kR2Analyzer) {
            const T f = fir(S1, p), d = S0[p - m];
            y[2 * i] = mul(add(f, d), scale);
            y[2 * i + 1] = mul(r2_sub(f, d), scale);
        } else if (MODE == kR2Synthesizer || MODE == kR2Interp) {
            y[2 * i] = mul(S0[p - m], scale);
            y[2 * i + 1] = mul(fir(S1, p), scale);
        } else {                                             // filter: unit = a pair of input samples (the last may be half)
            const T *A = c0 ? S1 : S0, *B = c0 ? S0 : S1;
            {
                const T yi = A[p - m], yq = fir(B, p - 1);
                y[4 * i] = mul(r2_half(add(yi, yq)), scale);
                y[4 * i + 1] = mul(r2_half(r2_sub(yi, yq)), scale);
            }
            if (2 * i + 1 < nx) {
                const T yi = B[p - m], yq = fir(A, p);
                y[4 * i + 2] = mul(r2_half(add(yi, yq)), scale);
                y[4 * i + 3] = mul(r2_half(r2_sub(yi, yq)), scale);
            }
        }
    }
}

// the two windows after the block: last 2m samples of (window ++ the stream's new samples); n0 / n1 = new samples of w0 / w1
template <class T, int MODE>
__global__ void resamp2_state_kernel(const T *__restrict__ state, const T *__restrict__ x, size_t nx, int m, int c0,
                                     long long n0, long long n1, T *__restrict__ state_next) {
    const int m2 = 2 * m, j = threadIdx.x + blockIdx.x * blockDim.x;
    if (j >= 2 * m2) return;
    const int s = j >= m2, k = j - s * m2;
    const long long u = (s ? n1 : n0) - m2 + k;
    state_next[j] = u < 0 ? state[s * m2 + (int)(m2 + u)] : r2_stream<T, MODE>(x, nx, s, u, c0);
}

template <class T, class C, int MODE>
static int launch_resamp2_mode(const T *state, const T *x, size_t nx, const C *h1, int m, C scale, int c0, T *y,
                               T *state_next, hipStream_t st) {
    size_t nunits, n0, n1;
    if (MODE == kR2Interp) { nunits = n0 = n1 = nx; }
    else if (MODE == kR2Filter) {
        nunits = (nx + 1) / 2;
        const size_t ne = (nx + 1) / 2, no = nx / 2;     // even / odd samples of the block
        n0 = c0 ? no : ne;
        n1 = c0 ? ne : no;
    } else {
        if (nx & 1) return fail(YAGI_ERR_CONFIG, "resamp2: this form consumes pairs of samples");
        nunits = n0 = n1 = nx / 2;
    }
    if (nunits == 0) return YAGI_OK;
    const size_t tiles = (nunits + kR2Tile - 1) / kR2Tile;
    if (tiles > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const size_t lds = 2 * (size_t)(kR2Tile + 2 * m) * sizeof(T);
    resamp2_kernel<T, C, MODE><<<(unsigned)tiles, kR2Lanes, lds, st>>>(state, x, nx, h1, m, scale, c0, y, nunits);
    YG_LAUNCH_CHECK();
    resamp2_state_kernel<T, MODE><<<(4 * m + 255) / 256, 256, 0, st>>>(state, x, nx, m, c0, (long long)n0, (long long)n1,
                                                                        state_next);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

template <class T, class C>
int launch_resamp2(int mode, const T *state, const T *x, size_t nx, const C *h1, int m, C scale, int toggle, T *y,
                   T *state_next, hipStream_t st) {
    if (m < 2 || m > kR2MaxSemiLen) return fail(YAGI_ERR_CONFIG, "resamp2: filter semi-length %d out of range", m);
    switch (mode) {
    case kR2Filter: return launch_resamp2_mode<T, C, kR2Filter>(state, x, nx, h1, m, scale, toggle & 1, y, state_next, st);
    case kR2Analyzer: return launch_resamp2_mode<T, C, kR2Analyzer>(state, x, nx, h1, m, scale, 0, y, state_next, st);
    case kR2Synthesizer: return launch_resamp2_mode<T, C, kR2Synthesizer>(state, x, nx, h1, m, scale, 0, y, state_next, st);
    case kR2Decim: return launch_resamp2_mode<T, C, kR2Decim>(state, x, nx, h1, m, scale, 0, y, state_next, st);
    case kR2Interp: return launch_resamp2_mode<T, C, kR2Interp>(state, x, nx, h1, m, scale, 0, y, state_next, st);
    default: return fail(YAGI_ERR_CONFIG, "resamp2: unknown form %d", mode);
    }
}

template int launch_resamp2<float, float>(int, const float *, const float *, size_t, const float *, int, float, int,
                                          float *, float *, hipStream_t);
template int launch_resamp2<cf32, float>(int, const cf32 *, const cf32 *, size_t, const float *, int, float, int, cf32 *,
                                         cf32 *, hipStream_t);
template int launch_resamp2<cf32, cf32>(int, const cf32 *, const cf32 *, size_t, const cf32 *, int, cf32, int, cf32 *,
                                        cf32 *, hipStream_t);

}  // namespace yagi
