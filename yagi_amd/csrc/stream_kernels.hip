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
#include "fft_core.hpp"
#include "kernels.hpp"

namespace yagi {

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
// zero-padded taps ever meet it).
__device__ __forceinline__ void stage_span(float2 *__restrict__ xs, const float2 *__restrict__ win,
                                           const float2 *__restrict__ x, long long base, int nspan,
                                           int L, long long x_len) {
    for (int u = threadIdx.x; u < nspan; u += 256) {
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
                          float2 *__restrict__ y, size_t ny) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    const int nrows = (kTile + Lp) >> 4;
    float2 *ys = xs + nrows * kRowPad;       // 4096 outputs, padded rows
    for (size_t tile = blockIdx.x; tile * kTile < ny; tile += gridDim.x) {
        const size_t o0 = tile * kTile;
        const long long base = (long long)o0 - (Lp - 1);
        stage_span(xs, win, x, base, kTile + Lp - 1, L, (long long)ny);
        __syncthreads();
        float2 acc[16];
        fir_tile_slide(acc, xs, taps_pad, Lp);
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
__global__ void __launch_bounds__(256)
firfft_crcf_4096_slide_kernel(const float2 *__restrict__ win, const float2 *__restrict__ x,
                              const float *__restrict__ taps_pad, int L, int Lp, float scale,
                              const float2 *__restrict__ tw, float2 *__restrict__ spectra,
                              size_t nframes) {
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    const int nrows = (kTile + Lp) >> 4;
    float2 *fl = xs + nrows * kRowPad;       // kFft4096LdsFloat2: FIR-output image, then FFT exchanges
    const long long x_len = (long long)nframes * kTile;
    for (size_t f = blockIdx.x; f < nframes; f += gridDim.x) {
        const long long base = (long long)f * kTile - (Lp - 1);
        stage_span(xs, win, x, base, kTile + Lp - 1, L, x_len);
        __syncthreads();
        float2 v[16];
        fir_tile_slide(v, xs, taps_pad, Lp);
        // lane t holds outputs 16t+j; FFT pass 1 wants lane b to hold outputs 256a+b
#pragma unroll
        for (int j = 0; j < 16; ++j)
            fl[threadIdx.x * kRowPad + j] = make_float2(v[j].x * scale, v[j].y * scale);
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = fl[padded(256 * a + threadIdx.x)];
        __syncthreads();
        fft4096_passes<-1>(v, fl, tw, spectra + f * kTile);
    }
}

static size_t slide_lds_bytes(int Lp) {
    const int nrows = (kTile + Lp) >> 4;
    return ((size_t)nrows * kRowPad + kFft4096LdsFloat2) * sizeof(float2);
}

static int raise_lds_limit(const void *fn, bool &done) {
    if (!done) {
        YG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        done = true;
    }
    return YAGI_OK;
}

// M = 1 crcf block FIR with the sliding kernel; taps_pad = h zero-padded to Lp = roundup(L, 32)
int launch_fir_crcf_slide(const cf32 *win, const cf32 *x, const float *taps_pad, int L, int Lp,
                          float scale, cf32 *y, size_t ny, hipStream_t st) {
    if (ny == 0) return YAGI_OK;
    if (Lp > kSlideMaxTaps || (Lp & 31)) return fail(YAGI_ERR_INTERNAL, "slide kernel: bad Lp %d", Lp);
    static bool raised = false;
    YG_TRY(raise_lds_limit(reinterpret_cast<const void *>(firfilt_crcf_slide_kernel), raised));
    size_t tiles = (ny + kTile - 1) / kTile;
    const unsigned grid = (unsigned)(tiles < 65536 ? tiles : 65536);
    firfilt_crcf_slide_kernel<<<grid, 256, slide_lds_bytes(Lp), st>>>(
        reinterpret_cast<const float2 *>(win), reinterpret_cast<const float2 *>(x), taps_pad, L, Lp,
        scale, reinterpret_cast<float2 *>(y), ny);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

int launch_firfft_crcf_4096(const cf32 *win, const cf32 *x, const float *taps_pad, const float *apack,
                            int L, int Lp, float scale, const cf32 *tw4096, cf32 *spectra,
                            size_t nframes, int variant, hipStream_t st) {
    (void)apack;
    (void)variant;
    if (nframes == 0) return YAGI_OK;
    if (Lp > kSlideMaxTaps || (Lp & 31)) return fail(YAGI_ERR_CONFIG, "fused stream: filter too long (%d taps)", L);
    static bool raised = false;
    YG_TRY(raise_lds_limit(reinterpret_cast<const void *>(firfft_crcf_4096_slide_kernel), raised));
    const unsigned grid = (unsigned)(nframes < 65536 ? nframes : 65536);
    firfft_crcf_4096_slide_kernel<<<grid, 256, slide_lds_bytes(Lp), st>>>(
        reinterpret_cast<const float2 *>(win), reinterpret_cast<const float2 *>(x), taps_pad, L, Lp,
        scale, reinterpret_cast<const float2 *>(tw4096), reinterpret_cast<float2 *>(spectra), nframes);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

}  // namespace yagi
