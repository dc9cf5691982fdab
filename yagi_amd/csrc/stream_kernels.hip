// stream_kernels.hip -- the hot case: FirFilter<Complex32,f32>::execute_block
// (src/filter/fir/firfilt.rs:267-278; inner product = src/dotprod/mod.rs:47-59) as a
// register-sliding direct-form kernel, standalone (config C2) and fused with the 4096-point
// forward FFT (headline stream).
//
// Tile = 4096 consecutive outputs per 256-lane workgroup, 16 CONSECUTIVE outputs per lane.
//   * input span (4096 + Lp - 1 samples, Lp = L rounded up to 32) is staged once in LDS in rows
//     of 16 samples padded to 17 (lane stride 34 dwords => every ds_read_b64 below touches all
//     64 banks exactly once);
//   * taps are wave-uniform: read through the scalar cache (s_load), used as SGPR operands;
//   * per block of 16 taps a lane loads ONE new row of 16 samples and reuses the previous row
//     from registers (31-sample sliding window), then issues 256 complex-by-real MACs
//     (512 v_fma_f32) -- 32 FMAs per LDS read, so the loop is FP32-VALU bound, which is the
//     roofline that bounds a direct-form 256-tap crcf filter on MI355X (64 flop/B).
// Algorithmic HBM traffic: 8 B read + 8 B written per sample (16 B/sample) in both kernels; in
// the fused kernel the FIR output goes registers -> LDS -> FFT registers and never touches HBM.
#include <algorithm>
#include <type_traits>

#include "fft_core.hpp"
#include "kernels.hpp"

namespace yagi {

constexpr bool kFusedTwp = true;           // twiddles by products also in the fused kernels (at 4 waves/SIMD: 0.174 -> 0.167 ms)
constexpr int kTile = 4096;                 // outputs per workgroup (= FFT length when fused)
constexpr int kRowPad = 17;                 // 16 samples + 1 pad


__device__ __forceinline__ int padded(int u) { return u + (u >> 4); }

// acc[j] += sum_{kk<16} h[kk] * W[j - kk + 15],  W = lo[0..16) ++ hi[0..15)
__device__ __forceinline__ void fir_block16(float2 (&acc)[16], const float2 (&lo)[16],
                                            const float2 (&hi)[16], const float *__restrict__ h) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const float hk = h[kk];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = j - kk + 15;
            const float2 w = (i < 16) ? lo[i] : hi[i - 16];
            acc[j].x = fmaf(w.x, hk, acc[j].x);
            acc[j].y = fmaf(w.y, hk, acc[j].y);
        }
    }
}

__device__ __forceinline__ void load_row(float2 (&w)[16], const float2 *__restrict__ xs, int row) {
    const float2 *p = xs + row * kRowPad;
#pragma unroll
    for (int e = 0; e < 16; ++e) w[e] = p[e];
}

// Stage X[base .. base + nspan) into the padded LDS image; X index < -L reads as zero (only the
// zero-padded taps ever meet it).  nspan = kTile + Lp (even), base = tile0 - (Lp - 1) (odd).
// Fast path (tile fully inside x, x 16-B aligned): every lane issues ALL its 16-byte loads first
// (9 x global_load_dwordx4 in flight per lane, 1 KiB contiguous per wave instruction), then writes
// LDS -- one HBM round trip per tile instead of one per element.
__device__ __forceinline__ void stage_span(float2 *__restrict__ xs, const float2 *__restrict__ win,
                                           const float2 *__restrict__ x, long long base, int nspan,
                                           int L, long long x_len) {
    const long long first = base - 1;                 // even X index: pairs (first + 2q, first + 2q + 1)
    const bool fast = first >= 0 && first + nspan + 2 <= x_len &&
                      ((reinterpret_cast<unsigned long long>(x) & 15ull) == 0);
    if (fast) {
        const float4 *src = reinterpret_cast<const float4 *>(x + first);
        const int npairs = (nspan + 2) >> 1;          // covers span indices -1 .. nspan
        // batches of 5 loads in flight per lane (bounded register footprint: 20 VGPRs)
        const int nthr = blockDim.x;
        for (int q0 = threadIdx.x; q0 < npairs; q0 += 5 * nthr) {
            float4 r[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int q = q0 + nthr * i;
                r[i] = src[q < npairs ? q : npairs - 1];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int q = q0 + nthr * i;
                const int u = 2 * q - 1;              // span index of r.xy ; r.zw is u + 1
                if (q < npairs) {
                    if (u >= 0) xs[padded(u)] = make_float2(r[i].x, r[i].y);
                    if (u + 1 < nspan) xs[padded(u + 1)] = make_float2(r[i].z, r[i].w);
                }
            }
        }
        return;
    }
    for (int u = threadIdx.x; u < nspan; u += blockDim.x) {
        const long long idx = base + u;
        float2 v = make_float2(0.f, 0.f);
        if (idx >= 0) { if (idx < x_len) v = x[idx]; }
        else if (idx >= -(long long)L) v = win[L + idx];
        xs[padded(u)] = v;
    }
}

// FIR of one tile: lane t ends with acc[j] = sum_k h[k] X[tile0 + 16t + j - k]  (unscaled)
__device__ __forceinline__ void fir_tile_slide(float2 (&acc)[16], const float2 *__restrict__ xs,
                                               const float *__restrict__ taps_pad, int Lp) {
    const int t = threadIdx.x;
    const int nb = Lp >> 4;                  // tap blocks (even)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = make_float2(0.f, 0.f);
    float2 wa[16], wb[16];
    load_row(wb, xs, t + nb);                // upper half for block 0
    for (int kb = 0; kb < nb; kb += 2) {
        load_row(wa, xs, t + nb - 1 - kb);
        fir_block16(acc, wa, wb, taps_pad + 16 * kb);
        load_row(wb, xs, t + nb - 2 - kb);
        fir_block16(acc, wb, wa, taps_pad + 16 * (kb + 1));
    }
}

// ---- standalone block FIR (config C2) ---------------------------------------------------------
__global__ void __launch_bounds__(256)
firfilt_crcf_slide_kernel(const float2 *__restrict__ win, const float2 *__restrict__ x,
                          const float *__restrict__ taps_pad, int L, int Lp, float scale,
                          float2 *__restrict__ y, size_t ny, float2 *__restrict__ win_next) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    float2 *ys = xs;                         // 4096 outputs, padded rows: reuses the span image once the FIR loop is done
    // the window after the block = last L samples of (win ++ x), written by the last workgroup (saves the
    // separate window-update launch: 4-5 us after every execute_block)
    if (win_next != nullptr && blockIdx.x == gridDim.x - 1)
        for (int j = threadIdx.x; j < L; j += 256) {
            const size_t c = ny + (size_t)j;
            win_next[j] = (c < (size_t)L) ? win[c] : x[c - (size_t)L];
        }
    // one tile per workgroup, no grid-stride loop (a loop makes the FFT twiddle / table loads loop-invariant
    // and LICM keeps them live across the FIR phase: +100 VGPRs in the fused kernels)
    {
        const size_t tile = blockIdx.x;
        const size_t o0 = tile * kTile;
        const long long base = (long long)o0 - (Lp - 1);
        stage_span(xs, win, x, base, kTile + Lp, L, (long long)ny);
        __syncthreads();
        float2 acc[16];
        fir_tile_slide(acc, xs, taps_pad, Lp);
        __syncthreads();                     // every lane is done reading the span
#pragma unroll
        for (int j = 0; j < 16; ++j)
            ys[threadIdx.x * kRowPad + j] = make_float2(acc[j].x * scale, acc[j].y * scale);
        __syncthreads();
        const int nt = (int)((ny - o0) < (size_t)kTile ? (ny - o0) : (size_t)kTile);
        for (int o = threadIdx.x; o < nt; o += 256) y[o0 + o] = ys[padded(o)];
        __syncthreads();
    }
}

// ---- fused FIR -> 4096-point forward FFT (the headline) -----------------------------------------
__global__ void __launch_bounds__(256, 2)
firfft_crcf_4096_slide_kernel(const float2 *__restrict__ win, const float2 *__restrict__ x,
                              const float *__restrict__ taps_pad, int L, int Lp, float scale,
                              const float2 *__restrict__ tw, float2 *__restrict__ spectra,
                              size_t nframes) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    float2 *fl = xs;                         // FIR-output image, then the FFT exchanges: reuses the span image
    const long long x_len = (long long)nframes * kTile;
    {
        const size_t f = blockIdx.x;        // one frame per workgroup, no grid-stride loop (see above)
        const long long base = (long long)f * kTile - (Lp - 1);
        stage_span(xs, win, x, base, kTile + Lp, L, x_len);
        __syncthreads();
        float2 v[16];
        fir_tile_slide(v, xs, taps_pad, Lp);
        __syncthreads();                     // every lane is done reading the span
        // lane t holds outputs 16t+j; FFT pass 1 wants lane b to hold outputs 256a+b
#pragma unroll
        for (int j = 0; j < 16; ++j)
            fl[threadIdx.x * kRowPad + j] = make_float2(v[j].x * scale, v[j].y * scale);
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = fl[padded(256 * a + threadIdx.x)];
        __syncthreads();
        fft4096_passes<-1, false, kFusedTwp>(v, fl, tw, spectra + f * kTile);
    }
}

static size_t slide_lds_bytes(int Lp) {
    const int nrows = (kTile + Lp) >> 4;     // span image; the 4352-float2 output/FFT image aliases it
    const size_t span = (size_t)nrows * kRowPad;
    return (span > (size_t)kFft4096LdsFloat2 ? span : (size_t)kFft4096LdsFloat2) * sizeof(float2);
}

static int raise_lds_limit(const void *fn, bool &done) {
    if (!done) {
        YG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        done = true;
    }
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// MFMA form of the same FIR: the batched filter re-cast as a dense tap x sample product.
//   D[m][n] += sum_u T[m][u] * X[u][n]      (v_mfma_f32_16x16x4_f32: exact f32 FMA chain)
//   m  = 16 consecutive output times (one row tile), n = 8 stream segments x {re, im},
//   T  = Toeplitz tap matrix T[m][u] = h[m - u + Lp - 1] (zero outside [0, L)),
//   X  = sample windows X[u][(seg,c)] = x[T0 + 64*seg - (Lp-1) + u].c
// A task = 512 consecutive outputs = 8 segments of 64 = 4 row tiles.  Shifting a row tile by 16
// outputs shifts its window by 16 samples = 4 K-steps, so ONE set of NS = ceil((Lp+15)/4) A
// registers (loaded once per kernel from the host-packed table `apack`) serves all four row tiles
// and every B register (one ds_read_b32 from the staged span) feeds up to 4 MFMAs.
// Overhead vs the minimum flop count: (Lp+15)/Lp (6 % at 256 taps).
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));


template <int NS>
__device__ __forceinline__ void load_apack(float (&a)[NS], const float *__restrict__ apack) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = apack[s * 64 + lane];
}

// acc[tt][r] = sum_k h[k] X[T0 + 64*seg + 16*tt + 4*(lane>>4) + r - k].c   (seg = (lane&15)>>1, c = lane&1)
template <int NS>
__device__ __forceinline__ void fir_task_mfma(f32x4 (&acc)[4], const float (&a)[NS],
                                              const float *__restrict__ xsf, int T0) {
    const int lane = threadIdx.x & 63;
    const int k = lane >> 4, j = lane & 15, seg = j >> 1, c = j & 1;
    // span index i = T0 + 64*seg + 4*sp + k ; padded(i) = i + (T0>>4) + 4*seg + (sp>>2)
    const float *p = xsf + 2 * (T0 + 64 * seg + k + (T0 >> 4) + 4 * seg) + c;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B operands are prefetched PF steps ahead (static ring in registers) and the instruction order
    // is pinned with sched_group_barrier (1 DS read, then the step's MFMAs) so the MFMA stream never
    // waits on an LDS round trip: the compiler then emits counted lgkmcnt(N) waits.
    constexpr int PF = 6;
    float bq[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) bq[i] = p[2 * (4 * i + (i >> 2))];
#pragma unroll
    for (int sp = 0; sp < NS + 12; ++sp) {
        const float b = bq[sp % PF];
        if (sp + PF < NS + 12) {
            bq[sp % PF] = p[2 * (4 * (sp + PF) + ((sp + PF) >> 2))];
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // 1 DS read
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int s = sp - 4 * tt;
            if (s >= 0 && s < NS) {
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b, acc[tt], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // then this MFMA
            }
        }
    }
}

// scatter a task's accumulators into the padded output image (float view)
__device__ __forceinline__ void store_task_mfma(const f32x4 (&acc)[4], float *__restrict__ outf, int T0,
                                                float scale) {
    const int lane = threadIdx.x & 63;
    const int k = lane >> 4, j = lane & 15, seg = j >> 1, c = j & 1;
    // o = T0 + 64*seg + 16*tt + 4*k + r ; padded(o) = o + (T0>>4) + 4*seg + tt
    float *p = outf + 2 * (T0 + 64 * seg + 4 * k + (T0 >> 4) + 4 * seg) + c;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[2 * (16 * tt + r + tt)] = acc[tt][r] * scale;
}

// A task's accumulators straight to global memory (the plain filter: no LDS image, no barriers).  Lane (k, seg, c)
// holds component c of the four consecutive outputs T0 + 64 seg + 16 tt + 4k + r; lanes j and j^1 (c = 0 / 1) hold
// the real and imaginary parts of the same four outputs.  One swap between the pair turns that into two whole
// complex samples per lane -- even lane: outputs r = 0, 1; odd lane: r = 2, 3 -- i.e. one 16-byte store per lane and
// row tile, 128 contiguous, line-aligned bytes per (seg, tt).  `nvalid` = outputs left from o_base (tail tile).
__device__ __forceinline__ void store_task_direct(const f32x4 (&acc)[4], float2 *__restrict__ out, size_t o_base,
                                                  long long nvalid, int T0, float scale) {
    const int lane = threadIdx.x & 63;
    const int k = lane >> 4, j = lane & 15, seg = j >> 1, c = j & 1;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const float v0 = acc[tt][0] * scale, v1 = acc[tt][1] * scale, v2 = acc[tt][2] * scale, v3 = acc[tt][3] * scale;
        // even lane sends (re2, re3), odd lane sends (im0, im1); quad_perm [1,0,3,2] swaps neighbours
        const float s0 = c ? v0 : v2, s1 = c ? v1 : v3;
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
        const float4 o = c ? make_float4(r0, v2, r1, v3) : make_float4(v0, r0, v1, r1);
        const int tm = T0 + 64 * seg + 16 * tt + 4 * k + 2 * c;              // first of the lane's two outputs
        if (tm + 1 < nvalid) *reinterpret_cast<float4 *>(out + o_base + tm) = o;
        else if (tm < nvalid) out[o_base + tm] = make_float2(o.x, o.y);
    }
}

// NW = waves per workgroup: 4 (two tasks per wave) or 8 (one task per wave, half the accumulators).
// TILE = outputs per workgroup: 4096 when FUSED (the frame); 2048 for the plain filter -- one task per wave, 16
// accumulator registers instead of 32, so four workgroups fit a CU and a 2^24-sample block is 8 full rounds of 1024
// workgroups (at 4096 outputs per workgroup three fit: 16 tiles per CU = 5 1/3 rounds, the last one a third full).
template <int NS, bool FUSED, int NW, int TILE>
__global__ void __launch_bounds__(64 * NW, (TILE == 2048 && NW == 4) ? 4 : 1)
fir_crcf_mfma_kernel(const float2 *__restrict__ win, const float2 *__restrict__ x,
                     const float *__restrict__ apack, int L, int Lp, float scale,
                     const float2 *__restrict__ tw, float2 *__restrict__ out, size_t n_units, bool direct,
                     float2 *__restrict__ win_next) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    const int wave = threadIdx.x >> 6;
    // the filter window after the block: the last L samples of (window ++ x), by the last workgroup (plain filter only)
    if (!FUSED && win_next != nullptr && blockIdx.x == gridDim.x - 1)
        for (int j = threadIdx.x; j < L; j += 64 * NW) {
            const size_t c = n_units + (size_t)j;
            win_next[j] = (c < (size_t)L) ? win[c] : x[c - (size_t)L];
        }
    static_assert(!FUSED || TILE == kTile, "the fused form transforms whole frames");
    constexpr int NT = TILE / 512 / NW;          // tasks per wave
    static_assert(NT >= 1 && NT * NW * 512 == TILE, "tile = NW x NT tasks of 512 outputs");
    // FUSED: n_units = frames (tile == frame); else n_units = output samples
    const size_t ntiles = FUSED ? n_units : (n_units + TILE - 1) / TILE;
    const long long x_len = FUSED ? (long long)n_units * TILE : (long long)n_units;
    {
        const size_t tile = blockIdx.x;     // one tile per workgroup, no grid-stride loop
        (void)ntiles;
        const size_t o0 = tile * TILE;
        stage_span(xs, win, x, (long long)o0 - (Lp - 1), TILE + Lp, L, x_len);
        // the Toeplitz tap registers are requested AFTER the span (vmcnt retires in order: in front of it they would
        // hold up the span's LDS writes) and are first needed behind the barrier
        float a[NS];
        load_apack<NS>(a, apack);
        __syncthreads();
        f32x4 acc[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            fir_task_mfma<NS>(acc[t], a, reinterpret_cast<const float *>(xs), 512 * (wave + NW * t));
        if (!FUSED && direct) {                  // plain filter, 16-byte aligned output: registers -> HBM
            const long long nvalid = (long long)(n_units - o0);
#pragma unroll
            for (int t = 0; t < NT; ++t) store_task_direct(acc[t], out, o0, nvalid, 512 * (wave + NW * t), scale);
            return;
        }
        __syncthreads();                         // all waves done reading the span: reuse it as output image
#pragma unroll
        for (int t = 0; t < NT; ++t)
            store_task_mfma(acc[t], reinterpret_cast<float *>(xs), 512 * (wave + NW * t), scale);
        __syncthreads();
        if (FUSED) {
            const bool active = (NW == 4) || threadIdx.x < 256;
            float2 v[16];
            if (active) {
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = xs[padded(256 * q + threadIdx.x)];
            }
            __syncthreads();
            fft4096_passes<-1, (NW > 4), kFusedTwp>(v, xs, tw, out + o0);
        } else {
            const int nt = (int)((n_units - o0) < (size_t)TILE ? (n_units - o0) : (size_t)TILE);
            for (int o = threadIdx.x; o < nt; o += 64 * NW) out[o0 + o] = xs[padded(o)];
            __syncthreads();
        }
    }
}

// host: A-operand table.  apack[s*64 + lane] = T[m = lane&15][u = 4s + (lane>>4)] = h[m - u + Lp - 1]
int mfma_lp_for(int L) { return L <= 64 ? 64 : (L <= 128 ? 128 : (L <= 256 ? 256 : 0)); }
size_t toeplitz_pack_floats(int Lp) { return (size_t)((Lp + 15 + 3) / 4) * 64; }
void pack_toeplitz_taps(const float *h, int L, int Lp, float *apack) {
    const int NS = (Lp + 15 + 3) / 4;
    for (int s = 0; s < NS; ++s)
        for (int lane = 0; lane < 64; ++lane) {
            const int m = lane & 15, u = 4 * s + (lane >> 4);
            const int k = m - u + Lp - 1;
            apack[s * 64 + lane] = (k >= 0 && k < L) ? h[k] : 0.0f;
        }
}

template <int NS, bool FUSED, int NW>
static int launch_mfma_t(const cf32 *win, const cf32 *x, const float *apack, int L, int Lp, float scale,
                         const cf32 *tw, cf32 *out, size_t n_units, hipStream_t st, cf32 *win_next) {
    constexpr int TILE = FUSED ? kTile : 2048;
    static bool raised = false;
    YG_TRY(raise_lds_limit(reinterpret_cast<const void *>(fir_crcf_mfma_kernel<NS, FUSED, NW, TILE>), raised));
    const size_t ntiles = FUSED ? n_units : (n_units + TILE - 1) / TILE;
    if (ntiles > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const unsigned grid = (unsigned)ntiles;
    const size_t lds = FUSED ? slide_lds_bytes(Lp) : (size_t)((TILE + Lp) >> 4) * kRowPad * sizeof(float2);
    fir_crcf_mfma_kernel<NS, FUSED, NW, TILE><<<grid, 64 * NW, lds, st>>>(
        reinterpret_cast<const float2 *>(win), reinterpret_cast<const float2 *>(x), apack, L, Lp, scale,
        reinterpret_cast<const float2 *>(tw), reinterpret_cast<float2 *>(out), n_units,
        !FUSED && (reinterpret_cast<unsigned long long>(out) & 15ull) == 0, reinterpret_cast<float2 *>(win_next));
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// The plain filter as a PERSISTENT matrix-pipe kernel: a workgroup walks the tiles b, b + G, b + 2G, ... of 2048
// outputs and keeps the pipe fed across tiles -- the 16-byte-per-step tap registers are loaded once per workgroup, the
// next tile's span goes from HBM straight into the OTHER span buffer while the current tile's MFMA phase runs (LDS-DMA
// buffer loads: no registers held, no LDS write instructions), and one barrier per tile hands the buffers over.  In the one-tile-per-workgroup form every wave stages, waits, computes and stores in turn and the pipe
// idles whenever the four waves of a SIMD are all outside their MFMA phase (84 % busy, profiles/r02_pmc_mfma.txt).
// Span layout: 2 float2 of padding per 64 samples, so the B operand of lane (seg, k, c) sits on bank
// (132 seg + 2 k + c) mod 32 -- the eight segments of a wave instruction on eight different banks (the shared
// 17/16 row layout of the sliding kernel put segments s and s + 4 on one bank: 47 % of this kernel's LDS cycles).
// ---------------------------------------------------------------------------------------------
constexpr int kMTile = 2048;
__host__ __device__ __forceinline__ constexpr int pad2(int i) { return i + 2 * (i >> 6); }

template <int NS>
__device__ __forceinline__ void fir_task_mfma2(f32x4 (&acc)[4], const float (&a)[NS],
                                               const float *__restrict__ xsf, int T0) {
    const int lane = threadIdx.x & 63;
    const int k = lane >> 4, j = lane & 15, seg = j >> 1, c = j & 1;
    // span index i = T0 + 64 seg + 4 sp + k ; pad2(i) = i + 2 (T0 >> 6) + 2 seg + 2 (sp >> 4)
    const float *p = xsf + 2 * (T0 + 2 * (T0 >> 6) + 66 * seg + k) + c;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PF = 6;
    float bq[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) bq[i] = p[2 * (4 * i + 2 * (i >> 4))];
#pragma unroll
    for (int sp = 0; sp < NS + 12; ++sp) {
        const float b = bq[sp % PF];
        if (sp + PF < NS + 12) {
            bq[sp % PF] = p[2 * (4 * (sp + PF) + 2 * ((sp + PF) >> 4))];
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // 1 DS read
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int s = sp - 4 * tt;
            if (s >= 0 && s < NS) {
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b, acc[tt], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // then this MFMA
            }
        }
    }
}

template <int NS>
__global__ void __launch_bounds__(256, 4)
fir_crcf_mfma_stream_kernel(const float2 *__restrict__ win, const float2 *__restrict__ x,
                            const float *__restrict__ apack, int L, float scale, float2 *__restrict__ out, size_t n,
                            float2 *__restrict__ win_next, unsigned ntiles) {
    constexpr int Lp = 4 * NS - 16;                    // 68 -> 256 taps, 36 -> 128, 20 -> 64
    constexpr int SPAN = kMTile + Lp;                  // samples a tile reads: outputs + Lp - 1 of history (+ 1)
    constexpr int NLD = (SPAN + 255) / 256;
    constexpr int NCH = ((SPAN + 31) / 32 + 3) / 4;    // 256-byte chunks per wave
    constexpr bool NCH4 = ((SPAN + 31) / 32) % 4 == 0;
    constexpr int BUF = pad2(SPAN + 64);
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6);       // scalar: M0 and the chunk offsets hang on it
    if (win_next != nullptr && blockIdx.x == gridDim.x - 1)
        for (int j = t; j < L; j += 256) {
            const size_t c = n + (size_t)j;
            win_next[j] = (c < (size_t)L) ? win[c] : x[c - (size_t)L];
        }
    unsigned tile = blockIdx.x;
    {   // first tile of the workgroup: the only one that can reach back into the window
        const long long base = (long long)tile * kMTile - (Lp - 1);
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int u = t + 256 * q;
            const long long idx = base + u;
            float2 v = make_float2(0.f, 0.f);
            if (idx >= 0) { if (idx < (long long)n) v = ld_stream(x + idx); }
            else if (idx >= -(long long)L) v = win[L + idx];
            if (u < SPAN) xs[pad2(u)] = v;
        }
    }
    float a[NS];
    load_apack<NS>(a, apack);
    __syncthreads();
    int cur = 0;
    const int lane = t & 63;
    for (;;) {
        const unsigned next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        if (has_next) {
            // the next tile's span straight from HBM into the other buffer (LDS-DMA: no registers held across the MFMA
            // phase): a wave instruction moves 256 contiguous bytes = half a 64-sample row; beyond n the descriptor reads
            // zero.  next >= 1, so the span starts inside x.
            const size_t s0 = (size_t)next * kMTile - (Lp - 1);
            const size_t avail = n - s0;
            const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + s0, (unsigned)(avail < (size_t)SPAN ? avail : (size_t)SPAN) * 8u);
            float2 *dst = xs + (cur ^ 1) * BUF;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = wave + 4 * i;            // chunk of 32 samples
                if (NCH4 || c < (SPAN + 31) / 32)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(dst + pad2(32 * c)), 4,
                                                             4 * lane, 256 * c, 0, kStreamLoad);
            }
        }
        f32x4 acc[4];
        fir_task_mfma2<NS>(acc, a, reinterpret_cast<const float *>(xs + cur * BUF), 512 * wave);
        __syncthreads();                               // the DMA has landed (vmcnt) and every wave is done with `cur`
        store_task_direct(acc, out, (size_t)tile * kMTile, (long long)(n - (size_t)tile * kMTile), 512 * wave, scale);
        if (!has_next) break;
        cur ^= 1;
        tile = next;
    }
}

template <int NS>
static int launch_mfma_stream(const cf32 *win, const cf32 *x, const float *apack, int L, float scale, cf32 *out, size_t n,
                              hipStream_t st, cf32 *win_next) {
    constexpr int Lp = 4 * NS - 16, SPAN = kMTile + Lp, BUF = pad2(SPAN + 64);
    const size_t ntiles = (n + kMTile - 1) / kMTile;
    if (ntiles > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t pr;
        YG_HIP(hipGetDevice(&dev));
        YG_HIP(hipGetDeviceProperties(&pr, dev));
        cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
        const unsigned grid = (unsigned)std::min<size_t>(ntiles, (size_t)cus * 4);       // four workgroups per CU stay resident (3: +0.9 %, 2: +2.8 %, 5: +13.5 %)
    fir_crcf_mfma_stream_kernel<NS><<<grid, 256, 2 * BUF * sizeof(float2), st>>>(
        reinterpret_cast<const float2 *>(win), reinterpret_cast<const float2 *>(x), apack, L, scale,
        reinterpret_cast<float2 *>(out), n, reinterpret_cast<float2 *>(win_next), (unsigned)ntiles);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

template <bool FUSED>
static int launch_mfma(const cf32 *win, const cf32 *x, const float *apack, int L, int Lp, float scale,
                       const cf32 *tw, cf32 *out, size_t n_units, hipStream_t st, cf32 *win_next = nullptr) {
    switch (Lp) {
        case 64: return launch_mfma_t<20, FUSED, 4>(win, x, apack, L, Lp, scale, tw, out, n_units, st, win_next);
        case 128: return launch_mfma_t<36, FUSED, 4>(win, x, apack, L, Lp, scale, tw, out, n_units, st, win_next);
        case 256: return launch_mfma_t<68, FUSED, 4>(win, x, apack, L, Lp, scale, tw, out, n_units, st, win_next);
    }
    return fail(YAGI_ERR_INTERNAL, "mfma FIR: unsupported padded length %d", Lp);
}

int launch_fir_crcf_mfma(const cf32 *win, const cf32 *x, const float *apack, int L, int Lp, float scale,
                         cf32 *y, size_t ny, hipStream_t st, cf32 *win_next) {
    if (ny == 0) return YAGI_OK;
    // blocks of at least one tile per resident workgroup, 16-byte aligned output: the persistent form
    if (ny >= ((size_t)1 << 21) && (reinterpret_cast<unsigned long long>(y) & 15ull) == 0 && ny < ((size_t)1 << 31)) {
        switch (Lp) {
            case 64: return launch_mfma_stream<20>(win, x, apack, L, scale, y, ny, st, win_next);
            case 128: return launch_mfma_stream<36>(win, x, apack, L, scale, y, ny, st, win_next);
            case 256: return launch_mfma_stream<68>(win, x, apack, L, scale, y, ny, st, win_next);
        }
    }
    return launch_mfma<false>(win, x, apack, L, Lp, scale, nullptr, y, ny, st, win_next);
}

// ---------------------------------------------------------------------------------------------
// Fast convolution (overlap-save) form of firfilt_crcf: the reference's own fast-convolution object is
// FftFilt (src/filter/fftfilt.rs:103-138, overlap-add); this is the overlap-save arrangement of the
// same identity, one 4096-point block per workgroup:
//   block b reads X[V*b - (L-1) .. V*b + V) (4096 samples, V = 4096 - (L-1) valid outputs),
//   FFT (3 register passes) -> the spectrum stays in registers in exactly the lane layout the next
//   transform's first pass wants (lane t holds bins t + 256d) -> multiply by FFT{h} * scale/4096 ->
//   inverse FFT -> the last V time samples are y[V*b .. V*b + V).
// 2 x 245 760 flop per 4096-V.. block instead of 4*L flop per sample: ~134 flop/sample at L = 256, so the
// kernel is bound by HBM/LDS, not FP32.  Traffic: 8*(4096/V) B read + 8 B written per sample.
// Results agree with the direct form to f32 rounding (rel 1e-6) but are NOT exact for integer inputs:
// FirFilter picks it only for long blocks (capi.hip), never for the per-sample / short-block calls.
//
// All three type combinations run on it (KIND):
//   crcf  complex samples, real taps, real scale
//   cccf  complex samples, complex taps (FFT{h} is simply not conjugate-symmetric), complex scale
//   rrrf  real samples: TWO consecutive blocks ride one complex transform as its real and imaginary parts
//         (h real => IFFT(H FFT(a + ib)) = h*a + i h*b), so a workgroup produces 2V real outputs
// ---------------------------------------------------------------------------------------------
// x[-pre .. x_avail) is readable (pre = samples of the same stream stored in front of x[0], used when a
// long block is processed in chunks); older samples come from `win` (the L-sample filter window).
// One block (rrrf: pair of blocks) per workgroup, one launch for all blocks.  Interior blocks (everything
// inside x) load straight into registers; the few boundary blocks (stream start / end) take a block-uniform
// branch that stages the block through the LDS buffer with per-sample checks.  (Boundary blocks in their own
// 1-block launches cost ~9 us each, 13 % of a step.)  No grid-stride loop: inside one, the twiddle and FFT{h}
// loads are loop-invariant and LICM keeps all 48 of them live across both transforms (256 VGPRs + spills
// vs 104).
enum { kConvCrcf = 0, kConvRrrf = 1, kConvCccf = 2 };

template <class T>
__device__ __forceinline__ T conv_fetch(const T *__restrict__ win, const T *__restrict__ x, long long idx,
                                        long long pre, long long x_avail, int L) {
    T s{};
    if (idx >= -pre) { if (idx < x_avail) s = x[idx]; }
    else if (idx >= -(long long)L) s = win[L + idx];
    return s;
}

template <int KIND>
__global__ void __launch_bounds__(256)
firfilt_fftconv_kernel(const void *__restrict__ win_, const void *__restrict__ x_, long long pre,
                       long long x_avail, const float2 *__restrict__ hs, float2 sc, int L, int V, int P0,
                       const float2 *__restrict__ tws,
                       void *__restrict__ y_, size_t ny, void *__restrict__ win_next_) {
    constexpr bool REAL = KIND == kConvRrrf;
    using T_ = typename std::conditional<REAL, float, float2>::type;
    const T_ *win = static_cast<const T_ *>(win_), *x = static_cast<const T_ *>(x_);
    T_ *y = static_cast<T_ *>(y_);
    // the last workgroup also writes the filter window after this call: the last L samples of win ++ x[0, x_avail)
    // (saves the separate update launch; win_next may be null)
    if (win_next_ && blockIdx.x == gridDim.x - 1) {
        T_ *wn = static_cast<T_ *>(win_next_);
        for (int j = threadIdx.x; j < L; j += 256) {
            const long long c = x_avail + j;                     // index into win ++ x
            wn[j] = (c < (long long)L) ? win[c] : x[c - L];
        }
    }
    __shared__ float2 lds[kFft4096LdsFloat2 + kFft4096TabFloat2];           // 36 992 B: four workgroups per CU
    float2 *T = lds + kFft4096LdsFloat2;
    const float2 *ax = tws + 4096;
    const size_t k0 = (REAL ? 2 : 1) * (size_t)blockIdx.x;            // first conv block of this workgroup
    const long long base = (long long)k0 * V - P0;      // P0 = 4096 - V >= L - 1 samples of history lead every block
    const long long last = base + (REAL ? V : 0) + 4096;              // end of the (second) block's input
    const bool interior = base >= -pre && last <= x_avail && (k0 + (REAL ? 2 : 1)) * (size_t)V <= ny;
    float2 v[16];
    if (interior) {
        const T_ *src = x + base;                    // block-uniform base (SGPRs) + 32-bit lane offset
#pragma unroll
        for (unsigned a = 0; a < 16; ++a) {
            if constexpr (REAL) v[a] = make_float2(ld_stream(src + 256u * a + threadIdx.x), ld_stream(src + V + 256u * a + threadIdx.x));
            else v[a] = ld_stream(src + 256u * a + threadIdx.x);
        }
        fft4096_tab_load(T, ax);
        __syncthreads();
    } else {
        for (int i = threadIdx.x; i < 4096; i += 256) {
            if constexpr (REAL)
                lds[i] = make_float2(conv_fetch(win, x, base + i, pre, x_avail, L),
                                     conv_fetch(win, x, base + V + i, pre, x_avail, L));
            else
                lds[i] = conv_fetch(win, x, base + i, pre, x_avail, L);
        }
        fft4096_tab_load(T, ax);
        __syncthreads();
#pragma unroll
        for (unsigned a = 0; a < 16; ++a) v[a] = lds[256u * a + threadIdx.x];
        __syncthreads();
    }
    // forward transform: pass 3 leaves bin t + 256 d in v[d], which is the pass-1 input layout again.  Twiddles as in
    // the headline kernel (fft_core.hpp fft4096_tab_to_regs): W_256 powers from a table in LDS, the W_4096 set from
    // four coalesced row entries + products -- the plain form gathered four table entries per lane and pass, twice per
    // transform, and the L1 / address path was what the kernel waited for
    fft4096_tab_to_regs<-1>(v, lds, T, ax);
    {
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(hs, 32768u);
        float2 hv[16];
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) hv[d] = buf_ld_aux<0>(rh, 8u * threadIdx.x, 2048u * d);     // a table: cached
#pragma unroll
        for (unsigned d = 0; d < 16; ++d) {
            const float2 p = cmul(v[d], hv[d]);
            if constexpr (KIND == kConvCccf) v[d] = cmul(p, sc);
            else v[d] = cscale(p, sc.x);
        }
    }
    fft4096_tab_to_regs<+1>(v, lds, T, ax);
    // time sample n = t + 256 d of the block; valid ones are n >= P0  ->  y[V*k + n - P0]
    const size_t o0 = k0 * (size_t)V;
    const size_t left = ny > o0 ? ny - o0 : 0;                         // outputs from o0 to the end of y
    const size_t lim0 = interior ? (size_t)V : (left < (size_t)V ? left : (size_t)V);
    T_ *yb = y + o0 - (size_t)P0;                     // yb[n], n = t + 256 d >= P0
#pragma unroll
    for (unsigned d = 0; d < 16; ++d) {
        const unsigned n = threadIdx.x + 256u * d;
        if (n < (unsigned)P0) continue;
        const size_t j = n - (unsigned)P0;
        if constexpr (REAL) {
            if (j < lim0) st_stream(yb + n, v[d].x);
            if (interior || j + (size_t)V < left) st_stream(yb + (size_t)V + n, v[d].y);
        } else {
            if (j < lim0) st_stream(yb + n, v[d]);
        }
    }
}

template <int KIND, class T>
static int launch_fir_fftconv_t(const T *win, const T *x, size_t pre, size_t x_avail, const cf32 *hs, cf32 scale,
                                int L, const cf32 *twf, const cf32 *twb, T *y, size_t ny, hipStream_t st, T *win_next) {
    if (ny == 0) return YAGI_OK;
    if (L < 1 || L > 2049) return fail(YAGI_ERR_CONFIG, "fast convolution kernel needs 1..2049 taps (got %d)", L);
    // outputs per block: what a 4096-point block leaves after the L - 1 samples of history, rounded down to whole
    // 128-byte lines, so that with line-aligned x and y every block's loads and stores are whole lines (V = 3840
    // instead of 3841 at 256 taps: the blocks started 8 bytes before a line and every wave access touched five lines)
    const int align = 128 / (int)sizeof(T);
    const int V = (4096 - (L - 1)) / align * align > 0 ? (4096 - (L - 1)) / align * align : 4096 - (L - 1);
    const int P0 = 4096 - V;
    const size_t nblk = (ny + V - 1) / V;
    const size_t nwg = KIND == kConvRrrf ? (nblk + 1) / 2 : nblk;
    if (nwg > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    firfilt_fftconv_kernel<KIND><<<(unsigned)nwg, 256, 0, st>>>(
        win, x, (long long)pre, (long long)x_avail, reinterpret_cast<const float2 *>(hs),
        make_float2(scale.re / 4096.0f, scale.im / 4096.0f), L, V, P0, reinterpret_cast<const float2 *>(twf), y, ny, win_next);
    (void)twb;
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

int launch_fir_crcf_fftconv(const cf32 *win, const cf32 *x, size_t pre, size_t x_avail, const cf32 *hs,
                            float scale, int L, const cf32 *twf, const cf32 *twb, cf32 *y, size_t ny,
                            hipStream_t st, cf32 *win_next) {
    return launch_fir_fftconv_t<kConvCrcf>(win, x, pre, x_avail, hs, cf32{scale, 0.f}, L, twf, twb, y, ny, st, win_next);
}
int launch_fir_cccf_fftconv(const cf32 *win, const cf32 *x, size_t pre, size_t x_avail, const cf32 *hs,
                            cf32 scale, int L, const cf32 *twf, const cf32 *twb, cf32 *y, size_t ny,
                            hipStream_t st, cf32 *win_next) {
    return launch_fir_fftconv_t<kConvCccf>(win, x, pre, x_avail, hs, scale, L, twf, twb, y, ny, st, win_next);
}
int launch_fir_rrrf_fftconv(const float *win, const float *x, size_t pre, size_t x_avail, const cf32 *hs,
                            float scale, int L, const cf32 *twf, const cf32 *twb, float *y, size_t ny,
                            hipStream_t st, float *win_next) {
    return launch_fir_fftconv_t<kConvRrrf>(win, x, pre, x_avail, hs, cf32{scale, 0.f}, L, twf, twb, y, ny, st, win_next);
}

// M = 1 crcf block FIR with the sliding kernel; taps_pad = h zero-padded to Lp = roundup(L, 32)
int launch_fir_crcf_slide(const cf32 *win, const cf32 *x, const float *taps_pad, int L, int Lp,
                          float scale, cf32 *y, size_t ny, hipStream_t st, cf32 *win_next) {
    if (ny == 0) return YAGI_OK;
    if (Lp > kSlideMaxTaps || (Lp & 31)) return fail(YAGI_ERR_INTERNAL, "slide kernel: bad Lp %d", Lp);
    static bool raised = false;
    YG_TRY(raise_lds_limit(reinterpret_cast<const void *>(firfilt_crcf_slide_kernel), raised));
    size_t tiles = (ny + kTile - 1) / kTile;
    if (tiles > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const unsigned grid = (unsigned)tiles;
    firfilt_crcf_slide_kernel<<<grid, 256, slide_lds_bytes(Lp), st>>>(
        reinterpret_cast<const float2 *>(win), reinterpret_cast<const float2 *>(x), taps_pad, L, Lp,
        scale, reinterpret_cast<float2 *>(y), ny, reinterpret_cast<float2 *>(win_next));
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

int launch_firfft_crcf_4096(const cf32 *win, const cf32 *x, const float *taps_pad, const float *apack,
                            int L, int Lp, int Lm, float scale, const cf32 *tw4096, cf32 *spectra,
                            size_t nframes, int variant, hipStream_t st) {
    if (nframes == 0) return YAGI_OK;
    if (Lp > kSlideMaxTaps || (Lp & 31)) return fail(YAGI_ERR_CONFIG, "fused stream: filter too long (%d taps)", L);
    if (variant == 2) {          // MFMA Toeplitz FIR
        if (!Lm || !apack) return fail(YAGI_ERR_CONFIG, "MFMA variant needs <= 256 taps (got %d)", L);
        return launch_mfma<true>(win, x, apack, L, Lm, scale, tw4096, spectra, nframes, st);
    }
    static bool raised = false;
    YG_TRY(raise_lds_limit(reinterpret_cast<const void *>(firfft_crcf_4096_slide_kernel), raised));
    if (nframes > 0x7fffffffull) return fail(YAGI_ERR_CONFIG, "block too large");
    const unsigned grid = (unsigned)nframes;
    firfft_crcf_4096_slide_kernel<<<grid, 256, slide_lds_bytes(Lp), st>>>(
        reinterpret_cast<const float2 *>(win), reinterpret_cast<const float2 *>(x), taps_pad, L, Lp,
        scale, reinterpret_cast<const float2 *>(tw4096), reinterpret_cast<float2 *>(spectra), nframes);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

}  // namespace yagi
