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

// sum_{k < m2} h1[k] w[k] in tap order (one FMA chain: bit-identical to the per-sample reference loop), the LDS reads of
// four taps issued together -- with one read per dependent FMA the loop ran at LDS latency (m2 is even and >= 4)
template <class T, class C>
__device__ __forceinline__ T r2_branch(const T *__restrict__ w, const C *__restrict__ h1, int m2) {
    T acc = zero_of<T>();
    int k = 0;
    for (; k + 4 <= m2; k += 4) {
        const T w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3];
        acc = mac(acc, w0, h1[k]);
        acc = mac(acc, w1, h1[k + 1]);
        acc = mac(acc, w2, h1[k + 2]);
        acc = mac(acc, w3, h1[k + 3]);
    }
    for (; k < m2; ++k) acc = mac(acc, w[k], h1[k]);
    return acc;
}

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
    auto fir = [&](const T *S, int q) { return r2_branch<T, C>(S + (q - (m2 - 1)), h1, m2); };
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

// ---------------------------------------------------------------------------------------------
// MsResamp2 decimator (msresamp2.rs:177-197), up to four half-band stages in ONE launch: a workgroup produces 256 final
// outputs; the stages run back to back on polyphase streams kept in LDS (stage k's outputs are written de-interleaved as
// stage k + 1's S1 / S0 streams), so only the block's input and the final outputs cross HBM -- the chained form moves
// 8 + 4 + 4 + 2 + 2 + 1 sample-sizes per input sample at three stages, this one 8 + 1.  Every stage evaluates exactly
// resamp2_kernel's decimator expression (same tap order), so the results are bit-identical to the chain.
// Stage k (processing order; k = 0 is the full-rate stage) owns outputs [ua_k, ub_k) and reads its streams at pair indices
// [ua_k - (2 m_k - 1), ub_k); indices below zero come from the stage's two windows (its object's state), which is also how
// a tile's halo is cut off at the start of the block; inside the block the halo is recomputed from the input.  The
// workgroup that owns the block's last output writes every stage's windows after the block from its LDS streams.
// ---------------------------------------------------------------------------------------------
constexpr int kMsMaxStages = 4, kMsTile = 256;
template <class T, class C>
struct MsDecimArgs {
    int ns;
    int m[kMsMaxStages];
    C scale[kMsMaxStages];
    const C *h1[kMsMaxStages];
    const T *state[kMsMaxStages];
    T *state_next[kMsMaxStages];
};

template <class T, class C, int S>
__global__ void __launch_bounds__(256)
msresamp2_decim_kernel(MsDecimArgs<T, C> a, const T *__restrict__ x, T *__restrict__ y, size_t nout, int tpw,
                       long long head_tiles, long long tail_first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ms_lds[];   // S is a template parameter: every per-stage array below lives in registers
    constexpr int NP = (1 << (S - 1)) + 1;                       // input pairs per lane and tile: (256 * 2^(S-1) + halo) / 256
    // the launch covers the outputs [0, 256 head_tiles) and [tail_first, nout): everything when tail_first = 256 head_tiles,
    // the two ends of the block when msresamp2_decim_fast_kernel takes the tiles in between
    const long long ntiles = head_tiles + (((long long)nout - tail_first) + kMsTile - 1) / kMsTile;
    const long long npairs = (long long)nout << (S - 1);         // pairs of the block's input
    const long long tile0 = (long long)blockIdx.x * tpw;
    const long long tend = tile0 + tpw < ntiles ? tile0 + tpw : ntiles;
    auto first_output = [&](long long tile) {
        return tile < head_tiles ? tile * kMsTile : tail_first + (tile - head_tiles) * kMsTile;
    };
    // output ranges of the stages of a tile, last to first (pair indices; may start below zero near the block's start)
    long long ua[kMsMaxStages], ub[kMsMaxStages];
    auto ranges = [&](long long tile) {
        const long long o0 = first_output(tile);
        const long long oend = tile < head_tiles && head_tiles * kMsTile < (long long)nout ? head_tiles * kMsTile : (long long)nout;
        long long lo = o0, hi = o0 + kMsTile < oend ? o0 + kMsTile : oend;
#pragma unroll
        for (int k = S - 1; k >= 0; --k) {
            ua[k] = lo;
            ub[k] = hi;
            lo = 2 * (lo - (2 * a.m[k] - 1));
            hi = 2 * hi;
        }
    };
    // the input pairs of a tile go through registers: the loads of tile i + 1 are in flight while tile i runs its stages
    // (a tile is ~20 KiB of input and five short barrier phases: one tile per workgroup left the memory system idle)
    T qe[NP], qo[NP];
    const int m20 = 2 * a.m[0];
    auto issue = [&](long long tile) {                           // clamped: entries below zero are replaced at commit
        long long lo = first_output(tile);
#pragma unroll
        for (int k = S - 1; k >= 1; --k) lo = 2 * (lo - (2 * a.m[k] - 1));
        const long long u0 = lo - (m20 - 1);                        // pair index of the tile's stream entry 0
        const long long ub0 = u0 > 0 ? u0 : 0;                       // first pair that exists
        const T *xb = x + 2 * ub0;                                   // uniform base; the lanes add 32-bit offsets
        const int shift = (int)(ub0 - u0), rmax = (int)(npairs - 1 - ub0);
#pragma unroll
        for (int b = 0; b < NP; ++b) {
            int r = (int)threadIdx.x + 256 * b - shift;              // entries below zero are replaced at commit,
            r = r < 0 ? 0 : (r < rmax ? r : rmax);                   // those past the block's end are never used
            qe[b] = xb[2 * r];
            qo[b] = xb[2 * r + 1];
        }
    };
    if (tile0 < tend) issue(tile0);
    for (long long tile = tile0; tile < tend; ++tile) {
        ranges(tile);
        // LDS: per stage two streams S0 / S1 of cnt_k = ub_k - ua_k + 2 m_k - 1 entries (entry j <-> pair index ua_k - (2m_k-1) + j)
        T *S0[kMsMaxStages], *S1[kMsMaxStages];
        {
            T *p = reinterpret_cast<T *>(ms_lds);
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const long long cnt = (ub[k] - ua[k]) + 2 * a.m[k] - 1;
                S0[k] = p;
                S1[k] = p + cnt;
                p += 2 * cnt;
            }
        }
        // stage 0's streams from the input (pairs: S1[u] = x[2u], S0[u] = x[2u+1]) or its windows
        {
            const long long u0 = ua[0] - (m20 - 1);
            const int cnt = (int)(ub[0] - u0);
            const int nneg = u0 < 0 ? (int)(-u0 < (long long)cnt ? -u0 : (long long)cnt) : 0;   // entries below index zero
            const int soff = u0 < 0 ? (int)(m20 + u0) : 0;           // window slot of entry 0 (may lie before the windows:
#pragma unroll                                                       // a halo reaches further back than they do near the
            for (int b = 0; b < NP; ++b) {                           // block's start; those entries are never used)
                const int j = (int)threadIdx.x + 256 * b;
                if (j < cnt) {
                    T e = qe[b], o = qo[b];
                    if (j < nneg) {
                        const bool held = soff + j >= 0;
                        o = held ? a.state[0][soff + j] : zero_of<T>();
                        e = held ? a.state[0][m20 + soff + j] : zero_of<T>();
                    }
                    S0[0][j] = o;
                    S1[0][j] = e;
                }
            }
        }
        if (tile + 1 < tend) issue(tile + 1);
#pragma unroll
        for (int k = 0; k < S; ++k) {
            __syncthreads();
            const int m = a.m[k], m2 = 2 * m;
            const C *__restrict__ h1 = a.h1[k];
            const C scale = a.scale[k];
            const long long u0 = ua[k] - (m2 - 1);                   // pair index of stream entry 0
            const bool last = k + 1 == S;
            // the next stage's streams below index zero: its windows
            long long v0 = 0;
            const int kn = k + 1 < S ? k + 1 : k;                    // next stage (unused when `last`; constant once unrolled)
            if (!last) {
                const int m2n = 2 * a.m[kn];
                v0 = ua[kn] - (m2n - 1);                              // pair index of the next stage's entry 0
                if (v0 < 0) {
                    const int nneg = (int)(-v0 < ub[kn] - v0 ? -v0 : ub[kn] - v0);
                    const int soff = (int)(m2n + v0);                 // window slot of entry 0 (see the stage-0 fill)
                    for (int j = threadIdx.x; j < nneg; j += 256) {
                        const bool held = soff + j >= 0;
                        S0[kn][j] = held ? a.state[kn][soff + j] : zero_of<T>();
                        S1[kn][j] = held ? a.state[kn][m2n + soff + j] : zero_of<T>();
                    }
                }
            }
            const long long first = ua[k] > 0 ? ua[k] : 0;
            const int nout_k = (int)(ub[k] - first);
            const int p0 = (int)(first - u0);                        // stream entry of the first output's pair index
            const int q0 = (int)(first - 2 * v0);                    // first output as a sample of the next stage's input, from its entry 0
            for (int j = threadIdx.x; j < nout_k; j += 256) {
                const int p = p0 + j;
                const T acc = r2_branch<T, C>(S1[k] + (p - (m2 - 1)), h1, m2);
                const T out = mul(add(S0[k][p - m], acc), scale);
                if (last) y[first + j] = out;
                else {
                    const int q = q0 + j;                             // sample index: pair q >> 1, odd -> S0, even -> S1
                    ((q & 1) ? S0[kn] : S1[kn])[q >> 1] = out;
                }
            }
        }
        // the windows after the block: last 2m pairs of every stage's streams (the tile that holds the block's end)
        if ((size_t)ub[S - 1] == nout) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int m2 = 2 * a.m[k];
                const int e0 = (int)(ub[k] - m2 - (ua[k] - (m2 - 1)));   // stream entry of pair ub_k - 2m (>= 0: the tile holds an output)
                for (int j = threadIdx.x; j < m2; j += 256) {
                    a.state_next[k][j] = S0[k][e0 + j];
                    a.state_next[k][m2 + j] = S1[k][e0 + j];
                }
            }
        }
        __syncthreads();                                             // the next tile overwrites the streams
    }
}

// ---------------------------------------------------------------------------------------------
// The same chain for the tiles in the middle of a block (S <= 3): every index of such a tile lies inside the block's
// input, its geometry is the same for all of them, and so nothing of it is computed per tile or per lane -- the general
// kernel above spends ~800 instructions per wave and tile (index arithmetic, range checks, one LDS read per tap and
// output) around ~100 of arithmetic and is bound by instruction issue (178 us of 228 with the loads removed, r03_notes.md).
// Here stage k gives every lane R_k = 2^(S-1-k) CONSECUTIVE outputs (4 / 2 / 1 at three stages: all stages keep the
// workgroup's lanes busy, which four-per-lane everywhere did not), the stream entries a lane needs are walked once with
// R_k accumulators (one LDS read per R_k taps' worth of FMAs), and the streams lie in R_k planes -- entry e at plane
// e mod R_k, position e div R_k -- so that for every step of the walk the lanes read consecutive positions of one plane,
// and a lane's R_k outputs land at position `lane` of the next stage's planes: no bank conflicts, no per-lane offsets.
// A tile makes F final outputs, F the largest count whose stage-0 streams fit 256 R_0 pairs (231 for m = 3 / 5 / 10).
// Each output is the same FMA chain in tap order as in resamp2_kernel: bit-identical to the chain of launches.
// ---------------------------------------------------------------------------------------------
struct MsFastGeom {
    int F;                      // final outputs per tile
    int n[3];                   // outputs of stage k per tile
    int cnt0;                   // input pairs per tile
    int U;                      // first input pair of a tile = 2^(S-1) * (its first output) - U
};

template <class T> struct MsPair { T e, o; };

// The walk over the entries of a lane's R outputs (R = 2, 4): entry t = R g + c feeds output r with tap t - r, so every
// output sees its taps in rising order.  With 2m = R Q + RHO (RHO = 0, or 2 at R = 4) the set of (c, r) pairs that are taps
// is fixed per group: g = 0: r <= c; 0 < g < Q: all; g = Q: r > c - RHO; g = Q + 1: r > c + R - RHO -- no run-time checks.
template <class T, class C, int R, int RHO>
__device__ __forceinline__ void ms_fast_walk(const T *__restrict__ S1l, int PS, const C *__restrict__ h, int m2, T (&acc)[R]) {
    const int Q = (m2 - RHO) / R;                                    // >= 1: 2m >= 4
    auto group = [&](int g, auto tap) {
        T wv[R];
#pragma unroll
        for (int c = 0; c < R; ++c) wv[c] = S1l[c * PS + g];
        const C *hg = h + R * g;
#pragma unroll
        for (int c = 0; c < R; ++c)
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (tap(c, r)) acc[r] = mac(acc[r], wv[c], hg[c - r]);
    };
    group(0, [](int c, int r) { return r <= c; });
    for (int g = 1; g < Q; ++g) group(g, [](int, int) { return true; });
    group(Q, [](int c, int r) { return r > c - RHO; });
    if constexpr (RHO >= 2) group(Q + 1, [](int c, int r) { return r > c + R - RHO; });
}

// R consecutive outputs of one stage for the lane whose first entry is at S1l / S0l (plane 0, the lane's position)
template <class T, class C, int R>
__device__ __forceinline__ void ms_fast_stage(const T *__restrict__ S1l, const T *__restrict__ S0l, int PS,
                                              const C *__restrict__ h, int m, C scale, T (&out)[R]) {
    constexpr int lgR = R == 1 ? 0 : (R == 2 ? 1 : 2);
    const int m2 = 2 * m;
    T acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = zero_of<T>();
    if constexpr (R == 1) {
        acc[0] = r2_branch<T, C>(S1l, h, m2);
    } else if constexpr (R == 2) {
        ms_fast_walk<T, C, 2, 0>(S1l, PS, h, m2, acc);
    } else {
        if (m2 & 2) ms_fast_walk<T, C, 4, 2>(S1l, PS, h, m2, acc);  // wave-uniform
        else ms_fast_walk<T, C, 4, 0>(S1l, PS, h, m2, acc);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int tt = r + m - 1;                                    // the delay branch: entry (first entry) + r + m - 1
        out[r] = mul(add(S0l[(tt & (R - 1)) * PS + (tt >> lgR)], acc[r]), scale);
    }
}

template <class T> __host__ __device__ constexpr int ms_plane_stride(int R) {
    // >= 256 positions; the stage-0 fill (lane -> plane lane mod R, position lane div R) spreads over the banks
    return 256 + (R > 1 ? (sizeof(T) == 4 ? 64 : 32) / R : 0);
}

template <class T, class C, int S>
__global__ void __launch_bounds__(256)
msresamp2_decim_fast_kernel(MsDecimArgs<T, C> a, MsFastGeom geo, const T *__restrict__ x, T *__restrict__ y,
                            long long o_first, int ntiles, int tpw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ms_lds[];
    static_assert(S >= 1 && S <= 3, "R_0 <= 4");
    constexpr int RL = S == 1 ? 4 : 1;                               // a single stage (a Resamp2 decimator's block) also runs four per lane
    constexpr int R0 = RL << (S - 1), lgR0 = S == 1 ? 2 : S - 1;
    const int tid = threadIdx.x;
    // streams: per stage S1 (filter branch), then S0 (delay branch), R_k planes each
    T *S1p[S], *S0p[S];
    T *Op = nullptr;                                                 // S = 1: the tile's outputs, turned into coalesced stores
    {
        T *p = reinterpret_cast<T *>(ms_lds);
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int R = RL << (S - 1 - k), PS = ms_plane_stride<T>(R);
            S1p[k] = p;
            S0p[k] = p + R * PS;
            p += 2 * R * PS;
        }
        if constexpr (S == 3) {                                      // stage 2's streams are written while stage 1 runs: stage 0's
            S1p[2] = S1p[0];                                         // are dead by then (25.6 KB instead of 29.7: six workgroups
            S0p[2] = S1p[0] + 256;                                   // per CU)
        }
        if constexpr (S == 1) Op = p;
    }
    const int tile0 = blockIdx.x * tpw;
    const int tend = tile0 + tpw < ntiles ? tile0 + tpw : ntiles;
    MsPair<T> q[R0];
    auto issue = [&](int tile) {
        const long long u0 = ((o_first + (long long)tile * geo.F) << (S - 1)) - geo.U;       // >= 0: the launcher's choice
        const MsPair<T> *xb = reinterpret_cast<const MsPair<T> *>(x) + u0;
#pragma unroll
        for (int b = 0; b < R0; ++b)
            if (tid + 256 * b < geo.cnt0) q[b] = xb[tid + 256 * b];
    };
    if (tile0 < tend) issue(tile0);
    for (int tile = tile0; tile < tend; ++tile) {
        {   // stage 0's streams: pair j = tid + 256 b -> plane j mod R0, position j div R0
            constexpr int PS = ms_plane_stride<T>(R0);
            T *d1 = S1p[0] + (tid & (R0 - 1)) * PS + (tid >> lgR0), *d0 = S0p[0] + (tid & (R0 - 1)) * PS + (tid >> lgR0);
#pragma unroll
            for (int b = 0; b < R0; ++b) {
                d1[(256 / R0) * b] = q[b].e;
                d0[(256 / R0) * b] = q[b].o;
            }
        }
        if (tile + 1 < tend) issue(tile + 1);
        auto lanes = [&](int k) { return (geo.n[k] + (1 << (S - 1 - k)) - 1) >> (S - 1 - k); };   // lanes with an output
        if constexpr (S == 3) {
            __syncthreads();
            if (tid < lanes(0)) {
                T out[4];
                ms_fast_stage<T, C, 4>(S1p[0] + tid, S0p[0] + tid, ms_plane_stride<T>(4), a.h1[0], a.m[0], a.scale[0], out);
                constexpr int PSn = ms_plane_stride<T>(2);           // outputs 4 tid + r = the next stage's pairs 2 tid, 2 tid + 1
                S1p[1][tid] = out[0];
                S0p[1][tid] = out[1];
                S1p[1][PSn + tid] = out[2];
                S0p[1][PSn + tid] = out[3];
            }
        }
        if constexpr (S >= 2) {
            constexpr int k = S - 2;
            __syncthreads();
            if (tid < lanes(k)) {
                T out[2];
                ms_fast_stage<T, C, 2>(S1p[k] + tid, S0p[k] + tid, ms_plane_stride<T>(2), a.h1[k], a.m[k], a.scale[k], out);
                S1p[k + 1][tid] = out[0];                            // outputs 2 tid, 2 tid + 1 = the next stage's pair tid
                S0p[k + 1][tid] = out[1];
            }
        }
        if constexpr (S == 1) {
            constexpr int PS = ms_plane_stride<T>(4);
            __syncthreads();
            if (4 * tid < geo.n[0]) {
                T out[4];
                ms_fast_stage<T, C, 4>(S1p[0] + tid, S0p[0] + tid, PS, a.h1[0], a.m[0], a.scale[0], out);
#pragma unroll
                for (int r = 0; r < 4; ++r) Op[r * PS + tid] = out[r];   // output 4 tid + r: plane r, position tid
            }
            __syncthreads();
            T *yt = y + o_first + (long long)tile * geo.F;
#pragma unroll
            for (int b = 0; b < 4; ++b) {                                // output j = tid + 256 b: plane j mod 4, position j div 4
                const int j = tid + 256 * b;
                if (j < geo.n[0]) yt[j] = Op[(tid & 3) * PS + (tid >> 2) + 64 * b];
            }
        } else {
            constexpr int k = S - 1;
            __syncthreads();
            if (tid < geo.n[k]) {
                T out[1];
                ms_fast_stage<T, C, 1>(S1p[k] + tid, S0p[k] + tid, 256, a.h1[k], a.m[k], a.scale[k], out);
                y[o_first + (long long)tile * geo.F + tid] = out[0];
            }
        }
        __syncthreads();                                             // the next tile overwrites the streams
    }
}

template <class T, class C, int S>
static void launch_ms_fast(const MsDecimArgs<T, C> &a, const MsFastGeom &geo, const T *x, T *y, long long o_first,
                           long long ntiles, hipStream_t st) {
    size_t lds = 0;
    constexpr int RL = S == 1 ? 4 : 1;
    for (int k = 0; k < (S == 3 ? 2 : S); ++k)                       // three stages: the last one's streams lie over stage 0's
        lds += 2 * (size_t)(RL << (S - 1 - k)) * ms_plane_stride<T>(RL << (S - 1 - k)) * sizeof(T);
    if (S == 1) lds += 4 * (size_t)ms_plane_stride<T>(4) * sizeof(T);   // the output planes
    lds += 64 * sizeof(T);                                           // a lane past the last output may read past its plane
    // two consecutive tiles per workgroup (the second tile's input in flight during the first): 1 / 2 / 4 / 8 / 16 tiles
    // measure 115.6 / 112.7 / 118.5 / 123.4 / 129.4 us at 2^26 inputs -- many short workgroups interleave better than
    // a few long ones prefetch
    const int tpw = S == 1 ? 1 : 2;                                  // one stage: tiles of ~1000 outputs, 1 / 2 / 4 per workgroup: 153.7 / 161.6 / 166.6 us
    const long long nblk = (ntiles + tpw - 1) / tpw;
    msresamp2_decim_fast_kernel<T, C, S><<<(unsigned)nblk, 256, lds, st>>>(a, geo, x, y, o_first, (int)ntiles, tpw);
}

template <class T, class C>
int launch_msresamp2_decim(int ns, const int *m, const C *scale, const C *const *h1, const T *const *state,
                           T *const *state_next, const T *x, T *y, size_t nout, hipStream_t st) {
    if (ns < 1 || ns > kMsMaxStages) return fail(YAGI_ERR_INTERNAL, "msresamp2: %d stages in one launch", ns);
    if (nout == 0) return YAGI_OK;
    MsDecimArgs<T, C> a;
    a.ns = ns;
    size_t lds = 0;
    for (int k = 0; k < ns; ++k) {
        a.m[k] = m[k];
        a.scale[k] = scale[k];
        a.h1[k] = h1[k];
        a.state[k] = state[k];
        a.state_next[k] = state_next[k];
    }
    {   // the widest tile: ranges of an interior workgroup
        long long lo = 0, hi = kMsTile;
        for (int k = ns - 1; k >= 0; --k) {
            lds += 2 * (size_t)((hi - lo) + 2 * m[k] - 1) * sizeof(T);
            lo = 2 * (lo - (2 * m[k] - 1));
            hi = 2 * hi;
        }
    }
    if (lds > 64 * 1024) return fail(YAGI_ERR_INTERNAL, "msresamp2: the fused chain needs %zu bytes of LDS", lds);
    // the middle of a long block goes to the fast kernel: tiles of F outputs from the first multiple of 256 whose halo
    // lies inside the block, the general kernel keeps that head and a tail of at least one output (it writes the windows)
    long long head_tiles = ((long long)nout + kMsTile - 1) / kMsTile, tail_first = head_tiles * kMsTile;
    MsFastGeom geo{};
    long long fast_tiles = 0;
    if (ns <= 3) {
        const int P2 = 1 << (ns - 1);                                 // input pairs per final output
        const int R0 = ns == 1 ? 4 : P2;                              // pairs per lane and tile
        for (int F = 256 * R0 / P2; F >= 64 && !geo.F; --F) {
            long long n = F, A = 0;
            int nk[3] = {0, 0, 0};
            for (int k = ns - 1; k >= 0; --k) {
                nk[k] = (int)n;
                if (k > 0) {
                    n = 2 * (n + 2 * m[k] - 1);
                    A = 2 * (A + 2 * m[k] - 1);
                }
            }
            const long long cnt0 = n + 2 * m[0] - 1;
            if (cnt0 <= 256 * R0) {
                geo.F = F;
                for (int k = 0; k < 3; ++k) geo.n[k] = nk[k];
                geo.cnt0 = (int)cnt0;
                geo.U = (int)(A + 2 * m[0] - 1);
            }
        }
        if (geo.F) {
            const long long o_min = (geo.U + P2 - 1) / P2;           // first output whose halo starts inside the block
            const long long head = (o_min + kMsTile - 1) / kMsTile;
            const long long room = (long long)nout - 1 - head * kMsTile;
            if (room >= 1024LL * 231) {                               // from ~2^18 outputs on
                fast_tiles = room / geo.F;
                head_tiles = head;
                tail_first = head * kMsTile + fast_tiles * geo.F;
            }
        }
    }
    const long long tiles = head_tiles + (((long long)nout - tail_first) + kMsTile - 1) / kMsTile;
    // consecutive tiles per workgroup (input prefetch across tiles) while >= ~4096 workgroups remain
    int tpw = 1;
    while (tpw < 8 && tiles / (2 * tpw) >= 4096) tpw *= 2;
    const long long nblk = (tiles + tpw - 1) / tpw;
    if (nblk > 0x7fffffffLL || fast_tiles > 0x7fffffffLL) return fail(YAGI_ERR_CONFIG, "block too large");
    if (fast_tiles) {
        const long long o_first = head_tiles * kMsTile;
        switch (ns) {
        case 1: launch_ms_fast<T, C, 1>(a, geo, x, y, o_first, fast_tiles, st); break;
        case 2: launch_ms_fast<T, C, 2>(a, geo, x, y, o_first, fast_tiles, st); break;
        default: launch_ms_fast<T, C, 3>(a, geo, x, y, o_first, fast_tiles, st); break;
        }
        YG_LAUNCH_CHECK();
    }
    switch (ns) {
    case 1: msresamp2_decim_kernel<T, C, 1><<<(unsigned)nblk, 256, lds, st>>>(a, x, y, nout, tpw, head_tiles, tail_first); break;
    case 2: msresamp2_decim_kernel<T, C, 2><<<(unsigned)nblk, 256, lds, st>>>(a, x, y, nout, tpw, head_tiles, tail_first); break;
    case 3: msresamp2_decim_kernel<T, C, 3><<<(unsigned)nblk, 256, lds, st>>>(a, x, y, nout, tpw, head_tiles, tail_first); break;
    default: msresamp2_decim_kernel<T, C, 4><<<(unsigned)nblk, 256, lds, st>>>(a, x, y, nout, tpw, head_tiles, tail_first); break;
    }
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
// ---------------------------------------------------------------------------------------------
// MsResamp2 interpolator (msresamp2.rs:154-175), up to four half-band stages in ONE launch: the mirror image of
// msresamp2_decim_kernel.  Level 0 is the block's input, stage k turns level k into level k + 1 (twice as long) with
// exactly resamp2_kernel's interpolator expression (y[2i] = S0[i - m] scale, y[2i+1] = scale sum_j h1[j] S1[i - (2m-1) + j];
// both streams carry the level's samples, below index zero the stage's two windows); the levels stay in LDS, the last
// stage stores to y.  A workgroup owns 256 input samples; a level's halo is recomputed from the level below, the
// workgroup that holds the block's end writes every stage's windows.  Bit-identical to the chain of Resamp2 stages.
// ---------------------------------------------------------------------------------------------
template <class T, class C, int S>
__global__ void __launch_bounds__(256)
msresamp2_interp_kernel(MsDecimArgs<T, C> a, const T *__restrict__ x, T *__restrict__ y, size_t nin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ms_lds[];
    // lo[k], hi[k]: range of level k this tile holds (level 0 = input); level S = the tile's outputs
    long long lo[kMsMaxStages + 1], hi[kMsMaxStages + 1];
    {
        const long long i0 = (long long)blockIdx.x * kMsTile;
        const long long i1 = i0 + kMsTile < (long long)nin ? i0 + kMsTile : (long long)nin;
        lo[S] = i0 << S;
        hi[S] = i1 << S;
#pragma unroll
        for (int k = S - 1; k >= 0; --k) {
            // stage k makes outputs [lo[k+1], hi[k+1]) (both even) from units [lo[k+1]/2, hi[k+1]/2): level-k samples from
            // (2 m_k - 1) before the first unit; stage k - 1 produces pairs, so the range is widened to even bounds
            long long l = (lo[k + 1] >> 1) - (2 * a.m[k] - 1), h = hi[k + 1] >> 1;
            if (k > 0) { l &= ~1ll; h = (h + 1) & ~1ll; }
            lo[k] = l;
            hi[k] = h;
        }
    }
    T *S0[kMsMaxStages], *S1[kMsMaxStages];                      // level k's two streams, entry j <-> index lo[k] + j
    {
        T *p = reinterpret_cast<T *>(ms_lds);
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const long long cnt = hi[k] - lo[k];
            S0[k] = p;
            S1[k] = p + cnt;
            p += 2 * cnt;
        }
    }
    // fill of a level's entries below index zero (the stage's windows) and, for level 0, the input itself
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int m2 = 2 * a.m[k];
        const int cnt = (int)(hi[k] - lo[k]);
        if (k == 0) {
            for (int j = threadIdx.x; j < cnt; j += 256) {
                const long long u = lo[0] + j;
                T v0, v1;
                if (u < 0) {                                     // a halo reaches further back than the stage's windows near
                    const bool held = u >= -m2;                  // the block's start: those entries are never used
                    v0 = held ? a.state[0][m2 + u] : zero_of<T>();
                    v1 = held ? a.state[0][m2 + m2 + u] : zero_of<T>();
                } else {
                    v0 = v1 = (u < (long long)nin) ? x[u] : zero_of<T>();
                }
                S0[0][j] = v0;
                S1[0][j] = v1;
            }
        } else if (lo[k] < 0) {
            const int nneg = (int)(-lo[k] < cnt ? -lo[k] : cnt);
            for (int j = threadIdx.x; j < nneg; j += 256) {
                const long long u = lo[k] + j;
                const bool held = u >= -m2;
                S0[k][j] = held ? a.state[k][m2 + u] : zero_of<T>();
                S1[k][j] = held ? a.state[k][m2 + m2 + u] : zero_of<T>();
            }
        }
    }
#pragma unroll
    for (int k = 0; k < S; ++k) {
        __syncthreads();
        const int m = a.m[k], m2 = 2 * m;
        const C *__restrict__ h1 = a.h1[k];
        const C scale = a.scale[k];
        const bool last = k + 1 == S;
        const int kn = k + 1 < S ? k + 1 : k;
        // units whose outputs the next level needs and that exist (index >= 0): [u_lo, u_hi)
        const long long first = (lo[k + 1] >> 1) > 0 ? (lo[k + 1] >> 1) : 0;
        const int nun = (int)((hi[k + 1] >> 1) - first);
        const int p0 = (int)(first - lo[k]);                         // entry of the first unit in level k's streams
        const int q0 = last ? 0 : (int)(2 * first - lo[kn]);         // entry of its first output in level k + 1
        for (int j = threadIdx.x; j < nun; j += 256) {
            const int p = p0 + j;
            const T acc = r2_branch<T, C>(S1[k] + (p - (m2 - 1)), h1, m2);
            const T e = mul(S0[k][p - m], scale), o = mul(acc, scale);
            if (last) {
                y[2 * (first + j)] = e;
                y[2 * (first + j) + 1] = o;
            } else {
                const int q = q0 + 2 * j;
                S0[kn][q] = e; S0[kn][q + 1] = o;
                S1[kn][q] = e; S1[kn][q + 1] = o;
            }
        }
    }
    // the windows after the block (both windows of a stage have seen the same samples since the block began; below index
    // zero they still hold what they held): the workgroup with the block's last input
    if ((size_t)(hi[S] >> S) == nin) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int m2 = 2 * a.m[k];
            const long long end = (long long)nin << k;               // samples of level k in the block
            for (int j = threadIdx.x; j < m2; j += 256) {
                const long long u = end - m2 + j;                    // >= lo[k]
                a.state_next[k][j] = S0[k][u - lo[k]];
                a.state_next[k][m2 + j] = S1[k][u - lo[k]];
            }
        }
    }
}

static size_t msresamp2_interp_lds_impl(int ns, const int *m, size_t elem) {
    long long lo = 0, hi = (long long)kMsTile << ns;
    size_t lds = 0;
    for (int k = ns - 1; k >= 0; --k) {
        long long l = (lo >> 1) - (2 * m[k] - 1), h = hi >> 1;
        if (k > 0) { l &= ~1ll; h = (h + 1) & ~1ll; }
        lds += 2 * (size_t)(h - l) * elem;
        lo = l;
        hi = h;
    }
    return lds;
}
size_t msresamp2_interp_lds(int ns, const int *m, size_t elem) { return msresamp2_interp_lds_impl(ns, m, elem); }

template <class T, class C>
int launch_msresamp2_interp(int ns, const int *m, const C *scale, const C *const *h1, const T *const *state,
                            T *const *state_next, const T *x, T *y, size_t nin, hipStream_t st) {
    if (ns < 1 || ns > kMsMaxStages) return fail(YAGI_ERR_INTERNAL, "msresamp2: %d stages in one launch", ns);
    if (nin == 0) return YAGI_OK;
    MsDecimArgs<T, C> a;
    a.ns = ns;
    for (int k = 0; k < ns; ++k) {
        a.m[k] = m[k];
        a.scale[k] = scale[k];
        a.h1[k] = h1[k];
        a.state[k] = state[k];
        a.state_next[k] = state_next[k];
    }
    const size_t lds = msresamp2_interp_lds_impl(ns, m, sizeof(T));
    if (lds > 64 * 1024) return fail(YAGI_ERR_INTERNAL, "msresamp2: the fused chain needs %zu bytes of LDS", lds);
    const size_t tiles = (nin + kMsTile - 1) / kMsTile;
    if (tiles > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    switch (ns) {
    case 1: msresamp2_interp_kernel<T, C, 1><<<(unsigned)tiles, 256, lds, st>>>(a, x, y, nin); break;
    case 2: msresamp2_interp_kernel<T, C, 2><<<(unsigned)tiles, 256, lds, st>>>(a, x, y, nin); break;
    case 3: msresamp2_interp_kernel<T, C, 3><<<(unsigned)tiles, 256, lds, st>>>(a, x, y, nin); break;
    default: msresamp2_interp_kernel<T, C, 4><<<(unsigned)tiles, 256, lds, st>>>(a, x, y, nin); break;
    }
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
template int launch_msresamp2_interp<float, float>(int, const int *, const float *, const float *const *, const float *const *,
                                                   float *const *, const float *, float *, size_t, hipStream_t);
template int launch_msresamp2_interp<cf32, float>(int, const int *, const float *, const float *const *, const cf32 *const *,
                                                  cf32 *const *, const cf32 *, cf32 *, size_t, hipStream_t);
template int launch_msresamp2_interp<cf32, cf32>(int, const int *, const cf32 *, const cf32 *const *, const cf32 *const *,
                                                 cf32 *const *, const cf32 *, cf32 *, size_t, hipStream_t);

// LDS the fused chain would need (host-side dispatch test)
size_t msresamp2_decim_lds(int ns, const int *m, size_t elem) {
    size_t lds = 0;
    long long lo = 0, hi = kMsTile;
    for (int k = ns - 1; k >= 0; --k) {
        lds += 2 * (size_t)((hi - lo) + 2 * m[k] - 1) * elem;
        lo = 2 * (lo - (2 * m[k] - 1));
        hi = 2 * hi;
    }
    return lds;
}

template int launch_msresamp2_decim<float, float>(int, const int *, const float *, const float *const *, const float *const *,
                                                  float *const *, const float *, float *, size_t, hipStream_t);
template int launch_msresamp2_decim<cf32, float>(int, const int *, const float *, const float *const *, const cf32 *const *,
                                                 cf32 *const *, const cf32 *, cf32 *, size_t, hipStream_t);
template int launch_msresamp2_decim<cf32, cf32>(int, const int *, const cf32 *, const cf32 *const *, const cf32 *const *,
                                                cf32 *const *, const cf32 *, cf32 *, size_t, hipStream_t);

template int launch_resamp2<float, float>(int, const float *, const float *, size_t, const float *, int, float, int,
                                          float *, float *, hipStream_t);
template int launch_resamp2<cf32, float>(int, const cf32 *, const cf32 *, size_t, const float *, int, float, int, cf32 *,
                                         cf32 *, hipStream_t);
template int launch_resamp2<cf32, cf32>(int, const cf32 *, const cf32 *, size_t, const cf32 *, int, cf32, int, cf32 *,
                                        cf32 *, hipStream_t);

}  // namespace yagi
