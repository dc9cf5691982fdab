// misc_kernels.hip -- synthetic-input generator, the standalone dotprod reduction kernel
// (trait DotProd, src/dotprod/mod.rs:13-73) and the filter-window update.
#include "devmath.hpp"
#include "kernels.hpp"

namespace yagi {

// ---------------------------------------------------------------------------------------------
// synthetic input: counter-based SplitMix64 + Box-Muller (shape of random/normal.rs:9-44)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <bool COMPLEX>
__global__ void __launch_bounds__(256) gen_kernel(uint64_t seed, uint64_t first, size_t n, float *out) {
    const float two_pi = 6.283185307179586f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t a = splitmix64_at(seed, 2 * (first + i));
        const uint64_t b = splitmix64_at(seed, 2 * (first + i) + 1);
        const float u1 = (float)((a >> 40) + 1) * (1.0f / 16777216.0f);
        const float u2 = (float)(b >> 40) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * logf(u1));
        const float th = two_pi * u2;
        if (COMPLEX) {
            float s, c;
            sincosf(th, &s, &c);
            reinterpret_cast<float2 *>(out)[i] = make_float2(r * 0.70710678f * c, r * 0.70710678f * s);
        } else {
            out[i] = r * sinf(th);
        }
    }
}

static int gen_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int launch_gen_real(uint64_t seed, uint64_t first, size_t n, float *x, hipStream_t st) {
    if (n == 0) return YAGI_OK;
    gen_kernel<false><<<gen_grid(n), 256, 0, st>>>(seed, first, n, x);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
int launch_gen_complex(uint64_t seed, uint64_t first, size_t n, cf32 *x, hipStream_t st) {
    if (n == 0) return YAGI_OK;
    gen_kernel<true><<<gen_grid(n), 256, 0, st>>>(seed, first, n, reinterpret_cast<float *>(x));
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

// ---------------------------------------------------------------------------------------------
// dotprod: per-lane strided FMA -> wave64 shuffle tree -> LDS across the 4 waves -> one value
// per workgroup; a second single-workgroup pass combines the per-workgroup partials in a fixed
// order, so the result does not depend on scheduling (no atomics).
// ---------------------------------------------------------------------------------------------
constexpr int kDotBlock = 256;
constexpr size_t kDotChunk = 256 * 64;      // elements per workgroup in pass 1

size_t dotprod_num_partials(size_t n) {
    size_t g = (n + kDotChunk - 1) / kDotChunk;
    if (g < 1) g = 1;
    if (g > 1024) g = 1024;
    return g;
}

template <class O>
__device__ __forceinline__ O block_reduce(O v, O *lds) {
    v = wave_reduce_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    O r = zero_of<O>();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 0; w < nw; ++w) r = add(r, lds[w]);
    }
    return r;       // valid in thread 0
}

template <class A, class B, class O, class S>
__global__ void __launch_bounds__(kDotBlock)
dotprod_pass1(const A *__restrict__ a, const B *__restrict__ b, size_t n, int rev_b, S post,
              O *__restrict__ out, int apply_post) {
    __shared__ O lds[kDotBlock / 64];
    O acc = zero_of<O>();
    size_t done = 0;
    // real x real, forward, on 16-byte boundaries: four elements per lane and trip through 16-byte loads (with
    // 4-byte loads rrrf reached 4.4 TB/s of operand reads, now 5.2; the complex kinds already load 8 bytes per
    // lane and measured no better with 16: 5.7-6.3 TB/s)
    constexpr int VEC = 4;
    if (sizeof(A) == 4 && sizeof(B) == 4 && !rev_b && ((reinterpret_cast<unsigned long long>(a) | reinterpret_cast<unsigned long long>(b)) & 15ull) == 0) {
        struct alignas(sizeof(A) * VEC) VA { A v[VEC]; };
        struct alignas(sizeof(B) * VEC) VB { B v[VEC]; };
        const VA *a4 = reinterpret_cast<const VA *>(a);
        const VB *b4 = reinterpret_cast<const VB *>(b);
        const size_t nv = n / VEC;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
            const VA av = a4[i];
            const VB bv = b4[i];
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc = mac(acc, av.v[c], bv.v[c]);
        }
        done = nv * VEC;
    }
    for (size_t i = done + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const B bv = rev_b ? b[n - 1 - i] : b[i];
        acc = mac(acc, a[i], bv);
    }
    O r = block_reduce(acc, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = apply_post ? mul(r, post) : r;
}

template <class O, class S>
__global__ void __launch_bounds__(kDotBlock)
dotprod_pass2(const O *__restrict__ partials, int np, S post, O *__restrict__ y) {
    __shared__ O lds[kDotBlock / 64];
    O acc = zero_of<O>();
    for (int i = threadIdx.x; i < np; i += blockDim.x) acc = add(acc, partials[i]);
    O r = block_reduce(acc, lds);
    if (threadIdx.x == 0) y[0] = mul(r, post);
}

template <class A, class B, class O, class S>
int launch_dotprod(const A *a, const B *b, size_t n, bool rev_b, S post, O *partials, O *y,
                   hipStream_t st) {
    const int g = (int)dotprod_num_partials(n);
    if (g == 1) {
        dotprod_pass1<A, B, O, S><<<1, kDotBlock, 0, st>>>(a, b, n, rev_b ? 1 : 0, post, y, 1);
        YG_LAUNCH_CHECK();
        return YAGI_OK;
    }
    dotprod_pass1<A, B, O, S><<<g, kDotBlock, 0, st>>>(a, b, n, rev_b ? 1 : 0, post, partials, 0);
    YG_LAUNCH_CHECK();
    dotprod_pass2<O, S><<<1, kDotBlock, 0, st>>>(partials, g, post, y);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}

template int launch_dotprod<float, float, float, float>(const float *, const float *, size_t, bool, float, float *, float *, hipStream_t);
template int launch_dotprod<float, cf32, cf32, float>(const float *, const cf32 *, size_t, bool, float, cf32 *, cf32 *, hipStream_t);
template int launch_dotprod<cf32, float, cf32, float>(const cf32 *, const float *, size_t, bool, float, cf32 *, cf32 *, hipStream_t);
template int launch_dotprod<cf32, cf32, cf32, float>(const cf32 *, const cf32 *, size_t, bool, float, cf32 *, cf32 *, hipStream_t);
template int launch_dotprod<cf32, cf32, cf32, cf32>(const cf32 *, const cf32 *, size_t, bool, cf32, cf32 *, cf32 *, hipStream_t);

// ---------------------------------------------------------------------------------------------
// window update: new_win = last L samples of (old_win ++ x[0..n))  (the state a Window<T> /
// VecDeque<T> holds after n pushes: window.rs:77-85, firfilt.rs:220-223)
// ---------------------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256)
update_window_kernel(const T *__restrict__ old_win, const T *__restrict__ x, size_t n, int L,
                     T *__restrict__ new_win) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < L; j += gridDim.x * blockDim.x) {
        const size_t c = n + (size_t)j;            // index into old_win ++ x
        new_win[j] = (c < (size_t)L) ? old_win[c] : x[c - (size_t)L];
    }
}

template <class T>
int launch_update_window(const T *old_win, const T *x, size_t n, int L, T *new_win, hipStream_t st) {
    int g = (L + 255) / 256;
    if (g > 64) g = 64;
    update_window_kernel<T><<<g, 256, 0, st>>>(old_win, x, n, L, new_win);
    YG_LAUNCH_CHECK();
    return YAGI_OK;
}
template int launch_update_window<float>(const float *, const float *, size_t, int, float *, hipStream_t);
template int launch_update_window<cf32>(const cf32 *, const cf32 *, size_t, int, cf32 *, hipStream_t);

}  // namespace yagi
