// devmath.hpp -- device-side scalar helpers: the three multiply-accumulate shapes of the FIR
// family (f32*f32, Complex*f32, Complex*Complex; num-complex semantics of
// src/dotprod/mod.rs:19-73 but with fused multiply-adds) and wave64 reductions.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace yagi {

template <class T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ float zero_of<float>() { return 0.0f; }
template <> __device__ __forceinline__ cf32 zero_of<cf32>() { return cf32{0.0f, 0.0f}; }

__device__ __forceinline__ float add(float a, float b) { return a + b; }
__device__ __forceinline__ cf32 add(cf32 a, cf32 b) { return cf32{a.re + b.re, a.im + b.im}; }

__device__ __forceinline__ float mul(float a, float b) { return a * b; }
__device__ __forceinline__ cf32 mul(cf32 a, float b) { return cf32{a.re * b, a.im * b}; }
__device__ __forceinline__ cf32 mul(float a, cf32 b) { return cf32{a * b.re, a * b.im}; }
__device__ __forceinline__ cf32 mul(cf32 a, cf32 b) {
    return cf32{fmaf(a.re, b.re, -a.im * b.im), fmaf(a.re, b.im, a.im * b.re)};
}

// acc + a*b
__device__ __forceinline__ float mac(float acc, float a, float b) { return fmaf(a, b, acc); }
__device__ __forceinline__ cf32 mac(cf32 acc, cf32 a, float b) {
    return cf32{fmaf(a.re, b, acc.re), fmaf(a.im, b, acc.im)};
}
__device__ __forceinline__ cf32 mac(cf32 acc, float a, cf32 b) {
    return cf32{fmaf(a, b.re, acc.re), fmaf(a, b.im, acc.im)};
}
__device__ __forceinline__ cf32 mac(cf32 acc, cf32 a, cf32 b) {
    return cf32{fmaf(a.re, b.re, fmaf(-a.im, b.im, acc.re)), fmaf(a.re, b.im, fmaf(a.im, b.re, acc.im))};
}

// wave64 butterfly sum: every lane ends with the total (xor tree: 32,16,8,4,2,1)
__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ cf32 wave_reduce_sum(cf32 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v.re += __shfl_xor(v.re, off, 64);
        v.im += __shfl_xor(v.im, off, 64);
    }
    return v;
}

// complex helpers on float2 (LDS / register FFT code works in float2)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward rotation) / +i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }

}  // namespace yagi
