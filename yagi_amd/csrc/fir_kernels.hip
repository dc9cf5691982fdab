// fir_kernels.hip -- direct-form FIR kernels for all three type combinations:
//   FirFilter::execute_block        (src/filter/fir/firfilt.rs:267-278)      M = 1
//   FirDecimationFilter::execute_block (src/filter/fir/firdecim.rs:179-205)  stride M
//   FirPfbFilter branch forms       (src/filter/fir/firpfb.rs:255-301)
//
// Data layout in HBM: samples contiguous (interleaved {re,im} for complex), taps h[0..L) in
// natural order, the filter state ("window") = the L samples preceding x[0], oldest first
// (what Window::read() returns, window.rs:66-68).  X below is the virtual stream win ++ x with
// X[0] = x[0]; the kernels never materialise it.
//
// fir_block_kernel: one workgroup produces `tile` consecutive outputs.  It stages the input
// span ((tile-1)*M + L samples, coalesced) and the taps in LDS, then every lane accumulates R
// outputs that are 256 apart, so at tap k the wave reads 64 consecutive samples (conflict-free
// ds_read) and one broadcast tap.  Algorithmic HBM traffic: sizeof(T)*(M + 1) bytes per output
// (+ the (L-1)-sample halo per tile, which is L2-resident).  This is the general kernel (any
// L, M, type); the crcf M=1 hot case has an MFMA version in fir_mfma.hip.
#include <type_traits>

#include "devmath.hpp"
#include <cstdlib>

#include "kernels.hpp"

namespace yagi {

constexpr int kFirBlock = 256;
constexpr int kFirR = 4;                         // outputs per lane
constexpr size_t kFirLdsBudget = 48 * 1024;

template <class T>
__device__ __forceinline__ T load_stream(const T *__restrict__ win, const T *__restrict__ x,
                                         long long idx, int L) {
    return (idx < 0) ? win[L + idx] : x[idx];
}

// The state a Window<T> holds after the block (window.rs:77-85): new_win = last L samples of (win ++ x[0..n)).
// Written by the LAST workgroup of the block's own kernel (reads only; `win_next` is the object's other window
// buffer), which saves the separate 4-5 us window-update launch after every execute_block.
template <class T>
__device__ __forceinline__ void write_next_window(const T *__restrict__ win, const T *__restrict__ x, size_t n, int L,
                                                  T *__restrict__ win_next) {
    if (win_next == nullptr || blockIdx.x != gridDim.x - 1) return;
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        const size_t c = n + (size_t)j;            // index into win ++ x
        win_next[j] = (c < (size_t)L) ? win[c] : x[c - (size_t)L];
    }
}

// STAGE = span in LDS.  The span is stored de-interleaved by decimation phase, xs[phase][j] = X[base + j*M + phase]
// (pitch P = ceil(span/M) + pad), so at tap k the 64 lanes of a wave -- 64 consecutive outputs, M samples
// apart in the stream -- read 64 CONSECUTIVE LDS words of one phase row (with the plain layout a decimator
// reads with a lane stride of M samples: 2M-way bank conflicts on ds_read_b64).  Taps come through the scalar
// cache (k is wave-uniform): one LDS read and one FMA group per MAC.
template <class K, bool STAGE>
__global__ void __launch_bounds__(kFirBlock)
fir_block_kernel(const typename K::T *__restrict__ win, const typename K::T *__restrict__ x,
                 const typename K::C *__restrict__ taps, int L, int M, typename K::C scale,
                 typename K::T *__restrict__ y, size_t ny, int tile, long long x_len,
                 typename K::T *__restrict__ win_next) {
    using T = typename K::T;
    using C = typename K::C;
    extern __shared__ __align__(16) unsigned char smem[];
    write_next_window(win, x, ny * (size_t)M, L, win_next);
    const size_t o0 = (size_t)blockIdx.x * (size_t)tile;
    const int nt = (int)((ny - o0) < (size_t)tile ? (ny - o0) : (size_t)tile);
    const long long base = (long long)o0 * M - (L - 1);
    const int span = (nt - 1) * M + L;
    const int pitch = (((tile - 1) * M + L + M - 1) / M) | 1;          // odd: phase rows start on different banks

    T *xs = reinterpret_cast<T *>(smem);
    if (STAGE) {
        // loads issued in batches before the LDS writes (devmath.hpp: batched_for)
        if (base >= 0 && base + span <= x_len) {       // block-uniform: the whole span lies inside x
            const T *src = x + base;
            if (M == 1) {
                batched_for<kFirBlock>(span, [&](int i) { return src[i]; }, [&](int i, T v) { xs[i] = v; });
            } else {
                batched_for<kFirBlock>(span, [&](int i) { return src[i]; }, [&](int i, T v) {
                    const int j = i / M, ph = i - j * M;
                    xs[ph * pitch + j] = v;
                });
            }
        } else {
            batched_for<kFirBlock>(span, [&](int i) { return load_stream(win, x, base + i, L); }, [&](int i, T v) {
                const int j = i / M, ph = i - j * M;
                xs[ph * pitch + j] = v;
            });
        }
        __syncthreads();
    }

    T acc[kFirR];
#pragma unroll
    for (int r = 0; r < kFirR; ++r) acc[r] = zero_of<T>();

    // newest sample of output o sits at span offset o*M + (L-1); tap k reads offset o*M + (L-1-k):
    // phase (L-1-k) mod M, row index o + (L-1-k) div M.  Taps are fetched 8 at a time (one scalar load),
    // a full tile (all 4 x 256 outputs exist) runs without per-output guards.
    auto run = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        int ph = (L - 1) % M, jq = (L - 1) / M;
        auto tap = [&](C hk, int k) {
            if (STAGE) {
                const T *row = xs + ph * pitch + jq + threadIdx.x;
#pragma unroll
                for (int r = 0; r < kFirR; ++r)
                    if (FULL || (int)threadIdx.x + r * kFirBlock < nt) acc[r] = mac(acc[r], row[r * kFirBlock], hk);
                if (--ph < 0) { ph = M - 1; --jq; }
            } else {
#pragma unroll
                for (int r = 0; r < kFirR; ++r) {
                    const int o = threadIdx.x + r * kFirBlock;
                    if (FULL || o < nt)
                        acc[r] = mac(acc[r], load_stream(win, x, base + (long long)o * M + (L - 1) - k, L), hk);
                }
            }
        };
        int k = 0;
        for (; k + 8 <= L; k += 8) {
            C hk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) hk[u] = taps[k + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) tap(hk[u], k + u);
        }
        for (; k < L; ++k) tap(taps[k], k);
    };
    if (nt == kFirR * kFirBlock) run(std::true_type{});
    else run(std::false_type{});
#pragma unroll
    for (int r = 0; r < kFirR; ++r) {
        const int o = threadIdx.x + r * kFirBlock;
        if (o < nt) y[o0 + o] = mul(acc[r], scale);
    }
}

// one chunk of eight samples of the rrrf body below (FULL: all eight inside d <= L-8; else wave-uniform guards)
template <bool FULL>
__device__ __forceinline__ void fir_consec_rrrf_chunk(v2f (&a2)[4], const float *xl, const float *__restrict__ taps,
                                                      int L, int Lp, int c) {
    const int d0 = 1 + 8 * c;                                // samples d0 .. d0+7 live in one pad group:
    const float *rb = xl + 9 * ((Lp >> 3) - c - 1);          // sample d0 + u at rb[7 - u]
    v2f g2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g2[i] = v2f{rb[2 * i], rb[2 * i + 1]};
    float ht[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) ht[i] = taps[(FULL || d0 + i < L) ? d0 + i : L - 1];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int gi = 7 - u;
        if (FULL || d0 + u <= L - 8) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const v2f hp = v2f{ht[u + 2 * p], ht[u + 2 * p + 1]};
                if (gi & 1)
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a2[p]) : "v"(g2[gi >> 1]), "s"(hp));
                else
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(a2[p]) : "v"(g2[gi >> 1]), "s"(hp));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rrrf body of fir_consec_kernel.  Real samples would have to sit in adjacent registers in exactly the pairing a
// packed FMA wants, and a sliding window breaks that on every other tap (the compiler patches it with one v_mov per
// two v_pk_fma_f32).  Turned around -- outer loop over the SAMPLE, which is broadcast to both halves (op_sel), packed
// operand = two adjacent TAPS from an SGPR pair -- no register pairing of samples is needed at all:
//     (y[8l+2p], y[8l+2p+1]) += X[8l - d] * (h[d+2p], h[d+2p+1]),   p < 4,  d = 1 .. L-8   (all 8 taps inside h)
// Samples come from LDS eight at a time (one pad group of the lane's row), taps through the scalar cache.  The
// 15 edge samples (d = -7..0 and d > L-8), where only some of the lane's outputs have a tap, are two static
// triangles of scalar FMAs.  Every output still adds its taps in the order k = 0, 1, .. L-1 with one FMA each: bit-identical
// to fir_block_kernel, and no tap outside h is ever multiplied.  xl = lane's row (slot of sample 8l - d:
// e + (e >> 3), e = Lp - d).  Needs L >= 16.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fir_consec_rrrf_body(const float *xl, const float *__restrict__ taps, int L, int Lp,
                                                     float (&acc)[8]) {
    v2f a2[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) a2[p] = v2f{0.f, 0.f};
    // head: samples X[8l + j], j = 7 .. 0 (d = -j); output r >= j takes tap h[r - j] -- a static triangle
    {
        const float *rb = xl + 9 * (Lp >> 3);                    // sample 8l + j at rb[j]
        float g[8], h0[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { g[j] = rb[j]; h0[j] = taps[j]; }
#pragma unroll
        for (int j = 7; j >= 0; --j)
#pragma unroll
            for (int r = j; r < 8; ++r) a2[r >> 1][r & 1] = fmaf(g[j], h0[r - j], a2[r >> 1][r & 1]);
    }
    // body: d = 1 .. L-8 in chunks of 8 samples (fir_consec_rrrf_chunk); the last chunk may be partial
    const int nch = (L - 8) >> 3;
#pragma unroll 1
    for (int c = 0; c < nch; ++c) fir_consec_rrrf_chunk<true>(a2, xl, taps, L, Lp, c);
    if (((L - 8) & 7) != 0) fir_consec_rrrf_chunk<false>(a2, xl, taps, L, Lp, nch);
    // tail: d = L-8+t, t = 1 .. 7; output r <= 7 - t takes tap h[L-8+t+r] -- the other static triangle
    {
        float hl[8];
#pragma unroll
        for (int i = 1; i < 8; ++i) hl[i] = taps[L - 8 + i];
#pragma unroll
        for (int t = 1; t < 8; ++t) {
            const int e = Lp - (L - 8 + t);
            const float xv = xl[e + (e >> 3)];
#pragma unroll
            for (int r = 0; r + t < 8; ++r) a2[r >> 1][r & 1] = fmaf(xv, hl[t + r], a2[r >> 1][r & 1]);
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = a2[r >> 1][r & 1];
}

// ---------------------------------------------------------------------------------------------
// M = 1 direct form with register reuse (all three type combinations): a lane owns 8 CONSECUTIVE outputs and walks
// the taps with an 8-sample register window that slides by one sample per tap -- one LDS read per 8 multiply-
// accumulates instead of one per MAC (the interleaved-output kernel above is bound by exactly that read).
// Tile = 256 lanes x 8 outputs; the span is staged with one pad element after every 8 samples, so that the 64 lanes
// of a wave (8 samples = 9 slots apart) hit distinct banks.  The window geometry is that of Lp = roundup(L, 8) taps
// (the span reaches up to 7 samples further back); the taps past L are skipped, never multiplied, so a NaN in the
// stream poisons exactly the outputs it poisons in the reference.  Taps come through the scalar cache 8 at a time.  Same sums in the same tap order as fir_block_kernel (bit-identical results).
// ---------------------------------------------------------------------------------------------
constexpr int kConsecR = 8, kConsecTile = 256 * kConsecR;
__device__ __forceinline__ int consec_pad(int i) { return i + (i >> 3); }

template <class K>
__global__ void __launch_bounds__(256)
fir_consec_kernel(const typename K::T *__restrict__ win, const typename K::T *__restrict__ x,
                  const typename K::C *__restrict__ taps, int L, typename K::C scale,
                  typename K::T *__restrict__ y, size_t ny, typename K::T *__restrict__ win_next) {
    using T = typename K::T;
    using C = typename K::C;
    extern __shared__ __align__(16) unsigned char smem[];
    T *xs = reinterpret_cast<T *>(smem);
    write_next_window(win, x, ny, L, win_next);
    const int Lp = (L + 7) & ~7;
    const size_t o0 = (size_t)blockIdx.x * kConsecTile;
    const int nt = (int)((ny - o0) < (size_t)kConsecTile ? (ny - o0) : (size_t)kConsecTile);
    const long long base = (long long)o0 - Lp;                  // stream index of span sample 0: a multiple of 8,
    const int span = kConsecTile + Lp;                          // so an aligned x is read 16 bytes per lane
    // X = win ++ x; indices before the window (no tap reaches them) and past the end of x read as zero
    if (base >= 0 && base + span <= (long long)ny && (reinterpret_cast<unsigned long long>(x) & 15ull) == 0) {
        constexpr int VEC = 16 / sizeof(T);
        const float4 *src = reinterpret_cast<const float4 *>(x + base);
        const int nvec = span / VEC;
        for (int v0 = threadIdx.x; v0 < nvec; v0 += 4 * 256) {  // four 16-byte loads in flight per lane
            float4 q[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) q[b] = src[v0 + 256 * b < nvec ? v0 + 256 * b : nvec - 1];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int v = v0 + 256 * b;
                if (v < nvec) {
                    T *dst = xs + consec_pad(v * VEC);           // VEC divides 8: the elements share a pad group
                    const T *e = reinterpret_cast<const T *>(&q[b]);
#pragma unroll
                    for (int c = 0; c < VEC; ++c) dst[c] = e[c];
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < span; i += 256) {
            const long long idx = base + i;
            T v = zero_of<T>();
            if (idx >= 0) { if (idx < (long long)ny) v = x[idx]; }
            else if (idx >= -(long long)L) v = win[L + idx];
            xs[consec_pad(i)] = v;
        }
    }
    __syncthreads();
    const int l = threadIdx.x;
    T acc[kConsecR], w[kConsecR];
#pragma unroll
    for (int r = 0; r < kConsecR; ++r) acc[r] = zero_of<T>();
    // at tap k (j = Lp-1-k) output r of the lane needs span sample 8l + q with q = r + j + 1, i.e. slot
    // 9l + q + (q >> 3); it is kept in w[(r + j) & 7]
    const T *xl = xs + 9 * l;
    bool done = false;
    if constexpr (K::id == 0) {
        if (L >= 16) {                                           // block-uniform
            fir_consec_rrrf_body(xl, taps, L, Lp, acc);
            done = true;
        }
    }
    if (!done) {
    {
        const int j = Lp - 1;                                    // j & 7 == 7
#pragma unroll
        for (int r = 1; r < kConsecR; ++r) w[(r + 7) & 7] = xl[(r + j + 1) + ((r + j + 1) >> 3)];
    }
    auto eight_taps = [&](const C (&hk)[8], int k0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int jb = Lp - 8 - k0;                              // multiple of 8: j = jb + 7 - u
        const T *xb = xl + jb + (jb >> 3);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            w[(7 - u) & 7] = xb[(8 - u) + ((8 - u) >> 3)];       // the window's new lowest sample (r = 0, q = j + 1)
            if (FULL || k0 + u < L) {                            // wave-uniform: taps past L are skipped, not zeroed
#pragma unroll
                for (int r = 0; r < kConsecR; ++r) acc[r] = mac(acc[r], w[(r + 7 - u) & 7], hk[u]);
            }
        }
    };
    int k0 = 0;
    for (; k0 + 8 <= L; k0 += 8) {
        C hk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hk[u] = taps[k0 + u];
        eight_taps(hk, k0, std::true_type{});
    }
    if (k0 < L) {
        C hk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hk[u] = taps[k0 + u < L ? k0 + u : L - 1];
        eight_taps(hk, k0, std::false_type{});
    }
    }
    const int o = kConsecR * l;
    if (o + kConsecR <= nt) {
#pragma unroll
        for (int r = 0; r < kConsecR; ++r) y[o0 + o + r] = mul(acc[r], scale);
    } else {
#pragma unroll
        for (int r = 0; r < kConsecR; ++r)
            if (o + r < nt) y[o0 + o + r] = mul(acc[r], scale);
    }
}

template <class K>
static int launch_fir_consec(const typename K::T *win, const typename K::T *x, const typename K::C *taps, int L,
                             typename K::C scale, typename K::T *y, size_t ny, hipStream_t st,
                             typename K::T *win_next) {
    using T = typename K::T;
    const int Lp = (L + 7) & ~7;
    const int span = kConsecTile + Lp;
    const size_t lds = (size_t)(span + (span >> 3) + 1) * sizeof(T);
    const size_t nblk = (ny + kConsecTile - 1) / kConsecTile;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    fir_consec_kernel<K><<<(unsigned)nblk, 256, lds, st>>>(win, x, taps, L, scale, y, ny, win_next);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// Decimator (M >= 2) with the same register window, one decimation phase at a time: the taps of phase p are
// h[L-1-p-M*i], i = 0.., and at step i output o reads row p of the phase-split span at index o + i -- so a lane
// that owns 8 consecutive outputs slides an 8-sample window along the row, one LDS read per 8 MACs.  Rows carry
// the 9/8 padding of fir_consec_kernel.  The sum runs phase by phase instead of in tap order (same products,
// f32 rounding may differ from fir_block_kernel in the last bit; exact on integer data).  NT lanes per
// workgroup (tile = 8*NT outputs) so that the M rows fit the LDS budget.
// ---------------------------------------------------------------------------------------------
template <class K, int NT, int R>
__global__ void __launch_bounds__(NT)
fir_decim_consec_kernel(const typename K::T *__restrict__ win, const typename K::T *__restrict__ x,
                        const typename K::C *__restrict__ taps, int L, int M, typename K::C scale,
                        typename K::T *__restrict__ y, size_t ny, int pitch, typename K::T *__restrict__ win_next) {
    using T = typename K::T;
    using C = typename K::C;
    constexpr int TILE = NT * R, LG = R == 8 ? 3 : 2;
    static_assert(R == 8 || R == 4, "window of 4 or 8 samples");
    extern __shared__ __align__(16) unsigned char smem[];
    T *xs = reinterpret_cast<T *>(smem);
    write_next_window(win, x, ny * (size_t)M, L, win_next);
    const size_t o0 = (size_t)blockIdx.x * TILE;
    const int nt = (int)((ny - o0) < (size_t)TILE ? (ny - o0) : (size_t)TILE);
    const long long base = (long long)o0 * M - (L - 1);         // stream index of row 0, entry 0
    const long long xlen = (long long)ny * M;
    const int ni = (L + M - 1) / M;                              // steps of the longest row
    const int total = (TILE + ni - 1) * M;                       // entries any (o, i) of a full tile can reach
    // entry e = jj*M + ph; (jj, ph) advance by NT entries per trip without a division
    const int djj = NT / M, dph = NT - djj * M;
    int jj = (int)threadIdx.x / M, ph = (int)threadIdx.x - jj * M;
    // eight loads per lane in flight before the LDS writes; (jj, ph) advance with the stores, in order
    const bool inside = base >= 0 && base + total <= xlen;       // block-uniform: every entry lies inside x
    // Interior tile, M a power of two with NT / M a multiple of R: the lane's decimation phase is fixed
    // (ph = tid mod M) and its row index advances by NT / M per trip, so the LDS slot advances by a constant and the
    // samples come through a buffer descriptor of exactly `total` entries (entries past it are range-checked away, never read):
    // no address arithmetic per entry -- the generic staging below spent as many vector instructions as the taps.
    const int lgM = 31 - __builtin_clz((unsigned)M);
    const bool desc = inside && (unsigned long long)total * sizeof(T) < 0xffffffffull;
    const bool fast = desc && (M & (M - 1)) == 0 && M <= NT && ((NT >> lgM) & (R - 1)) == 0;
    if (desc && !fast) {
        // any other M: the same descriptor loads, the (row, phase) of an entry advanced incrementally per trip
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + base, (unsigned)((size_t)total * sizeof(T)));
        const unsigned vo = (unsigned)sizeof(T) * threadIdx.x;
        for (int t0 = 0; t0 * NT < total; t0 += 8) {
            T r[8];
#pragma unroll
            for (int it = 0; it < 8; ++it)
                r[it] = buf_ld_t<T>(rx, vo + (unsigned)sizeof(T) * NT * (unsigned)(t0 + it), 0u);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if ((int)threadIdx.x + NT * (t0 + it) < total) xs[ph * pitch + jj + (jj >> LG)] = r[it];
                jj += djj; ph += dph;
                if (ph >= M) { ph -= M; ++jj; }
            }
        }
    } else if (fast) {
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + base, (unsigned)((size_t)total * sizeof(T)));
        const unsigned vo = (unsigned)sizeof(T) * threadIdx.x;
        const int rows = NT >> lgM;                              // row entries per trip
        const int sstride = rows + (rows >> LG);
        const int jj0 = (int)threadIdx.x >> lgM;
        T *dst = xs + ((int)threadIdx.x & (M - 1)) * pitch + jj0 + (jj0 >> LG);
        for (int t0 = 0; t0 * NT < total; t0 += 8) {
            T r[8];
#pragma unroll
            for (int it = 0; it < 8; ++it)       // the trip goes into the VGPR offset: that is the part the range check covers
                r[it] = buf_ld_t<T>(rx, vo + (unsigned)sizeof(T) * NT * (unsigned)(t0 + it), 0u);
            if ((t0 + 8) * NT <= total) {
#pragma unroll
                for (int it = 0; it < 8; ++it) dst[sstride * (t0 + it)] = r[it];
            } else {
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    if ((int)threadIdx.x + NT * (t0 + it) < total) dst[sstride * (t0 + it)] = r[it];
            }
        }
    } else
    for (int e0 = threadIdx.x; e0 < total; e0 += 8 * NT) {
        T r[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = e0 + NT * it;
            const long long idx = base + (e < total ? e : total - 1);
            T v = zero_of<T>();
            if (inside) v = x[idx];
            else if (idx < 0) v = win[L + idx];
            else if (idx < xlen) v = x[idx];
            r[it] = v;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (e0 + NT * it < total) xs[ph * pitch + jj + (jj >> LG)] = r[it];
            jj += djj; ph += dph;
            if (ph >= M) { ph -= M; ++jj; }
        }
    }
    __syncthreads();
    const int l = threadIdx.x;
    T acc[R], w[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = zero_of<T>();
    const int np_phases = M < L ? M : L;
#pragma unroll 1
    for (int ph = 0; ph < np_phases; ++ph) {
        const int n_p = (L - 1 - ph) / M + 1;
        const T *row = xs + ph * pitch + (R + 1) * l;            // slot of row index R l + q: (R+1) l + q + (q >> LG)
        const C *tp = taps + (L - 1 - ph);                       // step i multiplies by tp[-M*i]
#pragma unroll
        for (int r = 0; r < R - 1; ++r) w[r] = row[r];           // w[(r + i) & (R-1)] = row entry R l + r + i
        auto eight_steps = [&](int i0, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            C hk[R];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int i = (FULL || i0 + u < n_p) ? i0 + u : n_p - 1;
                hk[u] = tp[-(long long)M * i];
            }
            const T *rb = row + i0 + (i0 >> LG);                 // i0 is a multiple of R
#pragma unroll
            for (int u = 0; u < R; ++u) {
                w[(R - 1 + u) & (R - 1)] = rb[(R - 1 + u) + ((R - 1 + u) >> LG)];   // the window's new highest sample
                if (FULL || i0 + u < n_p) {
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = mac(acc[r], w[(r + u) & (R - 1)], hk[u]);
                }
            }
        };
        int i0 = 0;
        for (; i0 + R <= n_p; i0 += R) eight_steps(i0, std::true_type{});
        if (i0 < n_p) eight_steps(i0, std::false_type{});
    }
    const int o = R * l;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (o + r < nt) y[o0 + o + r] = mul(acc[r], scale);
}

// row pitch of the decimator kernel: entries any step can touch (tile + steps rounded up to 8, + the 8 of the
// last window refill), padded 9/8, odd so that the M rows start on different banks
static inline int decim_consec_pitch(int tile, int L, int M, int R) {
    const int ni = (L + M - 1) / M;
    const int n = tile + ((ni + R - 1) & ~(R - 1)) + R;
    return (n + n / R + 1) | 1;
}

template <class K, int NT, int R>
static int launch_fir_decim_consec(const typename K::T *win, const typename K::T *x, const typename K::C *taps, int L,
                                   int M, typename K::C scale, typename K::T *y, size_t ny, hipStream_t st,
                                   typename K::T *win_next) {
    using T = typename K::T;
    constexpr int TILE = NT * R;
    const int pitch = decim_consec_pitch(TILE, L, M, R);
    const size_t lds = (size_t)M * pitch * sizeof(T);
    const size_t nblk = (ny + TILE - 1) / TILE;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    fir_decim_consec_kernel<K, NT, R><<<(unsigned)nblk, NT, lds, st>>>(win, x, taps, L, M, scale, y, ny, pitch, win_next);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

template <class K>
int launch_fir_block(const typename K::T *win, const typename K::T *x, const typename K::C *taps,
                     int L, int M, typename K::C scale, typename K::T *y, size_t ny, hipStream_t st,
                     size_t x_len, typename K::T *win_next) {
    using T = typename K::T;
    if (ny == 0) return YAGI_OK;
    if (L <= 0 || M <= 0) return fail(YAGI_ERR_INTERNAL, "fir_block: bad L/M");
    // M = 1 with a block long enough to fill tiles, span within 48 KiB: the register-window kernel
    if (M == 1 && ny >= 512 && ((size_t)(kConsecTile + L + 8) * 9 / 8 + 1) * sizeof(T) <= kFirLdsBudget && x_len == 0)
        return launch_fir_consec<K>(win, x, taps, L, scale, y, ny, st, win_next);
    // decimators: the same register window per decimation phase, with the widest workgroup whose M rows fit
    // (pays once a phase has enough taps to amortise its window fill: 8-sample window from 32 taps per phase,
    // 4-sample window with full 256-lane workgroups from 8; YAGI_HIP_DECIM_WINDOW_MIN_STEPS overrides the 8)
    static const int min_steps = [] {
        const char *e = getenv("YAGI_HIP_DECIM_WINDOW_MIN_STEPS");
        return e ? atoi(e) : 8;
    }();
    if (M >= 2 && L >= M && L / M >= min_steps && ny >= 512 && x_len == 0) {
        auto fits = [&](int nt, int r) {
            return (size_t)M * decim_consec_pitch(nt * r, L, M, r) * sizeof(T) <= kFirLdsBudget;
        };
        const bool long_phase = L / M >= 32;
        // short phases: measured wins for complex samples and for M <= 4 (rrrf M = 8, 16 taps per phase: the general
        // kernel's 4-byte LDS reads are cheaper than eight window fills)
        const bool short_ok = sizeof(T) == 8 || M <= 4;
        if (long_phase && fits(256, 8)) return launch_fir_decim_consec<K, 256, 8>(win, x, taps, L, M, scale, y, ny, st, win_next);
        if (long_phase || short_ok) {
            if (fits(256, 4)) return launch_fir_decim_consec<K, 256, 4>(win, x, taps, L, M, scale, y, ny, st, win_next);
            if (fits(128, 4)) return launch_fir_decim_consec<K, 128, 4>(win, x, taps, L, M, scale, y, ny, st, win_next);
        }
        if (long_phase && fits(64, 8)) return launch_fir_decim_consec<K, 64, 8>(win, x, taps, L, M, scale, y, ny, st, win_next);
        if (long_phase && fits(64, 4)) return launch_fir_decim_consec<K, 64, 4>(win, x, taps, L, M, scale, y, ny, st, win_next);
    }
    // largest tile (<= R*256 outputs) whose phase-split span fits the LDS budget
    auto need = [&](int t) {
        const size_t pitch = (size_t)((((long long)(t - 1) * M + L + M - 1) / M) | 1);
        return (size_t)M * pitch * sizeof(T);
    };
    int tile = kFirR * kFirBlock;
    bool stage = true;
    while (tile >= 64 && need(tile) > kFirLdsBudget) tile /= 2;
    size_t lds = 0;
    if (tile < 64) {           // span too large for LDS staging: stream from L2/HBM
        stage = false;
        tile = kFirR * kFirBlock;
    } else {
        lds = need(tile);
    }
    const size_t nblk = (ny + tile - 1) / tile;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    // x_len = samples readable at x (0 = unknown: the outputs' own span, ny*M)
    const long long xl = (long long)(x_len ? x_len : ny * (size_t)M);
    if (stage)
        fir_block_kernel<K, true><<<(unsigned)nblk, kFirBlock, lds, st>>>(win, x, taps, L, M, scale, y, ny, tile, xl, win_next);
    else
        fir_block_kernel<K, false><<<(unsigned)nblk, kFirBlock, lds, st>>>(win, x, taps, L, M, scale, y, ny, tile, xl, win_next);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

template int launch_fir_block<RRRF>(const float *, const float *, const float *, int, int, float, float *, size_t, hipStream_t, size_t, float *);
template int launch_fir_block<CRCF>(const cf32 *, const cf32 *, const float *, int, int, float, cf32 *, size_t, hipStream_t, size_t, cf32 *);
template int launch_fir_block<CCCF>(const cf32 *, const cf32 *, const cf32 *, int, int, cf32, cf32 *, size_t, hipStream_t, size_t, cf32 *);

// ---------------------------------------------------------------------------------------------
// polyphase bank, all branches per pushed sample (interpolator form):
//   y[n*nf + i] = scale * sum_{k<Ls} hb[i][k] * X[n-k]
// A workgroup owns TN consecutive input samples x all nf branches.  Samples + halo are staged
// in LDS; branch taps are staged TRANSPOSED (hsT[k][i]) so the 64 lanes of a wave, which hold
// 64 consecutive branches i, read consecutive LDS words; stores are coalesced along i.
// Traffic: sizeof(T) read + nf*sizeof(T) written per input sample (write-bound).
// ---------------------------------------------------------------------------------------------
constexpr int kPfbTN = 64;

template <class K, bool TAPS_LDS>
__global__ void __launch_bounds__(256)
firpfb_all_kernel(const typename K::T *__restrict__ win, const typename K::T *__restrict__ x,
                  const typename K::C *__restrict__ hb, int nf, int Ls, typename K::C scale,
                  typename K::T *__restrict__ y, size_t n, typename K::T *__restrict__ win_next) {
    using T = typename K::T;
    using C = typename K::C;
    extern __shared__ __align__(16) unsigned char smem[];
    T *xs = reinterpret_cast<T *>(smem);                       // kPfbTN + Ls - 1 samples
    write_next_window(win, x, n, Ls, win_next);
    C *hsT = reinterpret_cast<C *>(smem + ((size_t)(kPfbTN + Ls - 1) * sizeof(T) + 15) / 16 * 16);
    const size_t n0 = (size_t)blockIdx.x * kPfbTN;
    const int nt = (int)((n - n0) < (size_t)kPfbTN ? (n - n0) : (size_t)kPfbTN);
    const long long base = (long long)n0 - (Ls - 1);
    batched_for<256>(nt + Ls - 1, [&](int i) { return load_stream(win, x, base + i, Ls); }, [&](int i, T v) { xs[i] = v; });
    if (TAPS_LDS)
        for (int e = threadIdx.x; e < nf * Ls; e += 256) {
            const int i = e / Ls, k = e - i * Ls;
            hsT[k * nf + i] = hb[e];
        }
    __syncthreads();
    const int total = nt * nf;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int nl = e / nf, i = e - nl * nf;
        T acc = zero_of<T>();
        for (int k = 0; k < Ls; ++k) {
            const C hk = TAPS_LDS ? hsT[k * nf + i] : hb[i * Ls + k];
            acc = mac(acc, xs[nl + (Ls - 1) - k], hk);
        }
        y[(n0 + nl) * (size_t)nf + i] = mul(acc, scale);
    }
}

// Few branches (interpolators: nf = 2 .. 16): with a lane per (sample, branch) the 64 lanes of a wave share 64/nf
// samples and nf taps -- two LDS reads per MAC and a tile of only 64 samples.  Here a lane owns ONE input sample
// and walks the branches four at a time: one LDS read (consecutive across lanes) per 4 MACs, the taps through
// the scalar cache (branch and tap index are wave-uniform), 256 samples per workgroup, and each lane stores 4
// consecutive outputs.  Same tap order as firpfb_all_kernel.
template <class K>
__global__ void __launch_bounds__(256)
firpfb_fewbranch_kernel(const typename K::T *__restrict__ win, const typename K::T *__restrict__ x,
                        const typename K::C *__restrict__ hb, int nf, int Ls, typename K::C scale,
                        typename K::T *__restrict__ y, size_t n, typename K::T *__restrict__ win_next) {
    using T = typename K::T;
    using C = typename K::C;
    extern __shared__ __align__(16) unsigned char smem[];
    T *xs = reinterpret_cast<T *>(smem);                       // 256 + Ls - 1 samples
    write_next_window(win, x, n, Ls, win_next);
    const size_t n0 = (size_t)blockIdx.x * 256;
    const int nt = (int)((n - n0) < (size_t)256 ? (n - n0) : (size_t)256);
    const long long base = (long long)n0 - (Ls - 1);
    batched_for<256>(nt + Ls - 1, [&](int i) { return load_stream(win, x, base + i, Ls); }, [&](int i, T v) { xs[i] = v; });
    __syncthreads();
    const int nl = threadIdx.x;
    if (nl >= nt) return;
    const T *xr = xs + nl + (Ls - 1);                          // tap k reads xr[-k]
    T *yo = y + (n0 + nl) * (size_t)nf;
    int g0 = 0;
    for (; g0 + 4 <= nf; g0 += 4) {
        const C *h0 = hb + (size_t)g0 * Ls;
        T acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = zero_of<T>();
#pragma unroll 4
        for (int k = 0; k < Ls; ++k) {
            const T sk = xr[-k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mac(acc[j], sk, h0[(size_t)j * Ls + k]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) yo[g0 + j] = mul(acc[j], scale);
    }
    for (; g0 < nf; ++g0) {                                    // the last nf % 4 branches
        const C *h0 = hb + (size_t)g0 * Ls;
        T acc = zero_of<T>();
        for (int k = 0; k < Ls; ++k) acc = mac(acc, xr[-k], h0[k]);
        yo[g0] = mul(acc, scale);
    }
}

template <class K>
int launch_firpfb_all(const typename K::T *win, const typename K::T *x, const typename K::C *hb,
                      int nf, int Ls, typename K::C scale, typename K::T *y, size_t n, hipStream_t st,
                      typename K::T *win_next) {
    using T = typename K::T;
    using C = typename K::C;
    if (n == 0) return YAGI_OK;
    if (nf <= 16 && (size_t)(256 + Ls - 1) * sizeof(T) <= kFirLdsBudget) {
        const size_t nblk = (n + 255) / 256;
        if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
        firpfb_fewbranch_kernel<K><<<(unsigned)nblk, 256, (size_t)(256 + Ls - 1) * sizeof(T), st>>>(win, x, hb, nf, Ls,
                                                                                                   scale, y, n, win_next);
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    }
    const size_t xs_bytes = ((size_t)(kPfbTN + Ls - 1) * sizeof(T) + 15) / 16 * 16;
    if (xs_bytes > kFirLdsBudget) return fail(YAGI_ERR_CONFIG, "branch filters too long (%d taps)", Ls);
    const size_t tap_bytes = (size_t)nf * Ls * sizeof(C);
    const bool taps_lds = xs_bytes + tap_bytes <= kFirLdsBudget;
    const size_t nblk = (n + kPfbTN - 1) / kPfbTN;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    if (taps_lds)
        firpfb_all_kernel<K, true><<<(unsigned)nblk, 256, xs_bytes + tap_bytes, st>>>(win, x, hb, nf, Ls, scale, y, n, win_next);
    else
        firpfb_all_kernel<K, false><<<(unsigned)nblk, 256, xs_bytes, st>>>(win, x, hb, nf, Ls, scale, y, n, win_next);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
template int launch_firpfb_all<RRRF>(const float *, const float *, const float *, int, int, float, float *, size_t, hipStream_t, float *);
template int launch_firpfb_all<CRCF>(const cf32 *, const cf32 *, const float *, int, int, float, cf32 *, size_t, hipStream_t, cf32 *);
template int launch_firpfb_all<CCCF>(const cf32 *, const cf32 *, const cf32 *, int, int, cf32, cf32 *, size_t, hipStream_t, cf32 *);

// ---------------------------------------------------------------------------------------------
// branch index per sample (the arbitrary resampler's access pattern, resamp.rs:141-154 with the
// phase schedule precomputed): y[n] = scale * sum_k hb[idx[n]][k] * X[n-k].
// One lane per output; samples from an LDS tile, taps gathered from L2.
// ---------------------------------------------------------------------------------------------
constexpr int kSelTile = 1024;

template <class K>
__global__ void __launch_bounds__(256)
firpfb_select_kernel(const typename K::T *__restrict__ win, const typename K::T *__restrict__ x,
                     const typename K::C *__restrict__ hb, const uint32_t *__restrict__ idx,
                     int nf, int Ls, typename K::C scale, typename K::T *__restrict__ y, size_t n) {
    using T = typename K::T;
    using C = typename K::C;
    extern __shared__ __align__(16) unsigned char smem[];
    T *xs = reinterpret_cast<T *>(smem);
    const size_t n0 = (size_t)blockIdx.x * kSelTile;
    const int nt = (int)((n - n0) < (size_t)kSelTile ? (n - n0) : (size_t)kSelTile);
    const long long base = (long long)n0 - (Ls - 1);
    batched_for<256>(nt + Ls - 1, [&](int i) { return load_stream(win, x, base + i, Ls); }, [&](int i, T v) { xs[i] = v; });
    __syncthreads();
    for (int nl = threadIdx.x; nl < nt; nl += 256) {
        uint32_t b = idx[n0 + nl];
        if (b >= (uint32_t)nf) b = (uint32_t)nf - 1;     // host validates; clamp keeps the read in range
        const C *hrow = hb + (size_t)b * Ls;
        T acc = zero_of<T>();
        for (int k = 0; k < Ls; ++k) acc = mac(acc, xs[nl + (Ls - 1) - k], hrow[k]);
        y[n0 + nl] = mul(acc, scale);
    }
}

template <class K>
int launch_firpfb_select(const typename K::T *win, const typename K::T *x, const typename K::C *hb,
                         const uint32_t *idx, int nf, int Ls, typename K::C scale,
                         typename K::T *y, size_t n, hipStream_t st) {
    using T = typename K::T;
    if (n == 0) return YAGI_OK;
    const size_t xs_bytes = (size_t)(kSelTile + Ls - 1) * sizeof(T);
    if (xs_bytes > kFirLdsBudget) return fail(YAGI_ERR_CONFIG, "branch filters too long (%d taps)", Ls);
    const size_t nblk = (n + kSelTile - 1) / kSelTile;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    firpfb_select_kernel<K><<<(unsigned)nblk, 256, xs_bytes, st>>>(win, x, hb, idx, nf, Ls, scale, y, n);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
template int launch_firpfb_select<RRRF>(const float *, const float *, const float *, const uint32_t *, int, int, float, float *, size_t, hipStream_t);
template int launch_firpfb_select<CRCF>(const cf32 *, const cf32 *, const float *, const uint32_t *, int, int, float, cf32 *, size_t, hipStream_t);
template int launch_firpfb_select<CCCF>(const cf32 *, const cf32 *, const cf32 *, const uint32_t *, int, int, cf32, cf32 *, size_t, hipStream_t);

// ---------------------------------------------------------------------------------------------
// Rresamp<T,Coeff>::execute_primitive over many blocks (rresamp.rs:162-183): every Q inputs give P
// outputs; output n of a block is branch (n Q) mod P of the bank, evaluated after input floor(n Q / P)
// of that block has been pushed -- a static schedule, so every output is an independent Ls-tap dot
// product:   y[blk P + n] = scale * sum_k hb[(nQ) mod P][k] X[blk Q + floor(nQ/P) - k].
// One lane per output; the tile's input span sits in LDS, the bank is gathered through L1/L2.
// ---------------------------------------------------------------------------------------------
template <class K>
__global__ void __launch_bounds__(256)
rresamp_kernel(const typename K::T *__restrict__ win, const typename K::T *__restrict__ x,
               const typename K::C *__restrict__ hb, int P, int Q, int Ls, typename K::C scale,
               typename K::T *__restrict__ y, size_t nblocks, int tile_blocks, typename K::T *__restrict__ win_next) {
    using T = typename K::T;
    using C = typename K::C;
    extern __shared__ __align__(16) unsigned char smem[];
    T *xs = reinterpret_cast<T *>(smem);
    write_next_window(win, x, nblocks * (size_t)Q, Ls, win_next);
    const size_t b0 = (size_t)blockIdx.x * tile_blocks;
    const int nb = (int)((nblocks - b0) < (size_t)tile_blocks ? (nblocks - b0) : (size_t)tile_blocks);
    const long long base = (long long)b0 * Q - (Ls - 1);
    const int span = nb * Q + Ls - 1;
    batched_for<256>(span, [&](int i) { return load_stream(win, x, base + i, Ls); }, [&](int i, T v) { xs[i] = v; });
    __syncthreads();
    const int nout = nb * P;
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int bl = o / P, n = o - bl * P;
        const int nq = n * Q, i = nq / P, br = nq - i * P;
        const C *hrow = hb + (size_t)br * Ls;
        const T *xp = xs + (Ls - 1) + bl * Q + i;
        // one FMA chain in tap order (bit-identical to the per-sample loop), the reads of four taps issued together
        T acc = zero_of<T>();
        int k = 0;
        for (; k + 4 <= Ls; k += 4) {
            const T x0 = xp[-k], x1 = xp[-k - 1], x2 = xp[-k - 2], x3 = xp[-k - 3];
            const C h0 = hrow[k], h1 = hrow[k + 1], h2 = hrow[k + 2], h3 = hrow[k + 3];
            acc = mac(acc, x0, h0);
            acc = mac(acc, x1, h1);
            acc = mac(acc, x2, h2);
            acc = mac(acc, x3, h3);
        }
        for (; k < Ls; ++k) acc = mac(acc, xp[-k], hrow[k]);
        y[b0 * P + o] = mul(acc, scale);
    }
}

template <class K>
int launch_rresamp(const typename K::T *win, const typename K::T *x, const typename K::C *hb, int P, int Q,
                   int Ls, typename K::C scale, typename K::T *y, size_t nblocks, hipStream_t st,
                   typename K::T *win_next) {
    using T = typename K::T;
    if (nblocks == 0) return YAGI_OK;
    // blocks per tile: ~1024 outputs or inputs, whichever is larger, inside the LDS budget
    const int big = P > Q ? P : Q;
    int tb = 1024 / big;
    if (tb < 1) tb = 1;
    auto bytes = [&](int t) { return ((size_t)t * Q + Ls - 1) * sizeof(T); };
    while (tb > 1 && bytes(tb) > kFirLdsBudget) tb /= 2;
    if (bytes(tb) > kFirLdsBudget) return fail(YAGI_ERR_CONFIG, "rresamp: Q and the branch length do not fit the LDS (%d, %d)", Q, Ls);
    const size_t nblk = (nblocks + tb - 1) / tb;
    if (nblk > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    rresamp_kernel<K><<<(unsigned)nblk, 256, bytes(tb), st>>>(win, x, hb, P, Q, Ls, scale, y, nblocks, tb, win_next);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
template int launch_rresamp<RRRF>(const float *, const float *, const float *, int, int, int, float, float *, size_t, hipStream_t, float *);
template int launch_rresamp<CRCF>(const cf32 *, const cf32 *, const float *, int, int, int, float, cf32 *, size_t, hipStream_t, cf32 *);
template int launch_rresamp<CCCF>(const cf32 *, const cf32 *, const cf32 *, int, int, int, cf32, cf32 *, size_t, hipStream_t, cf32 *);

}  // namespace yagi
