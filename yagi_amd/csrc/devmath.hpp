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

// complex helpers on float2 (LDS / register FFT code works in float2).
//
// A complex number is one 64-bit VGPR pair and every operation below is ONE packed instruction per
// pair of real operations (gfx950 v_pk_{add,mul,fma}_f32).  Left to itself the compiler gets the plain
// adds right but spends an extra v_mov + v_xor on every swizzled use -- the (re,im) swap of a complex
// product and of a multiplication by +-i -- where the packed encodings can select halves (op_sel) and
// negate per half (neg_lo / neg_hi) for free.  Those two shapes are therefore spelled out:
//   cmul        a*w      = pk_mul (a.x w.x, a.x w.y) ; pk_fma (a.y (-w.y) + ., a.y w.x + .)
//   addsub_rot  a +- (-+i) b   = two pk_adds reading b's halves crosswise
// (a 16-point butterfly + 15 twiddles: 110 VALU instructions instead of ~160).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 f2(v2f v) { return make_float2(v.x, v.y); }
__device__ __forceinline__ v2f tov(float2 a) { return v2f{a.x, a.y}; }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return f2(tov(a) + tov(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return f2(tov(a) - tov(b)); }
// a * w, w per lane
// (both instructions in ONE asm statement: between two statements hipcc pads a wait state -- an s_nop per product,
// ~8 % of a butterfly pass's issue slots; the hardware interlocks the dependent pair by itself)
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(tov(a)), "v"(tov(w)));
    return f2(r);
}
// a * conj(w)
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 w) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(r) : "v"(tov(a)), "v"(tov(w)));
    return f2(r);
}
template <int SIGN>
__device__ __forceinline__ float2 cmul_dir(float2 a, float2 w) { return SIGN < 0 ? cmul(a, w) : cmul_conj(a, w); }
// a * k, k wave-uniform (compile-time constants: lives in an SGPR pair)
__device__ __forceinline__ float2 cmul_k(float2 a, float2 k) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(tov(a)), "s"(tov(k)));
    return f2(r);
}
// a * s, s real
__device__ __forceinline__ float2 cscale(float2 a, float s) { return f2(tov(a) * s); }
// multiply by -i (forward rotation) / +i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }
// p = a + r b, m = a - r b with r = -i (SIGN < 0, forward transforms) or +i (SIGN > 0)
template <int SIGN>
__device__ __forceinline__ void addsub_rot(float2 a, float2 b, float2 &p, float2 &m) {
    v2f pp, mm;
    if (SIGN < 0) {      // r b = (b.y, -b.x)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(pp) : "v"(tov(a)), "v"(tov(b)));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(mm) : "v"(tov(a)), "v"(tov(b)));
    } else {             // r b = (-b.y, b.x)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(pp) : "v"(tov(a)), "v"(tov(b)));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(mm) : "v"(tov(a)), "v"(tov(b)));
    }
    p = f2(pp);
    m = f2(mm);
}

// Strided per-lane loop e = threadIdx.x, +NT, ... < total whose loads are ISSUED IN BATCHES OF EIGHT before the first
// store.  Written as a plain `for (e...) dst[f(e)] = src[e]` with a run-time bound, the compiler waits for every load
// before its store: one memory round trip per element and lane (measured: the LDS-staged FFT kernels went from
// 4.8 to 6.1 TB/s with nothing else changed).  `load(e)` must be valid for every e in [0, total).
template <int NT, int B, bool EXACT, class Load, class Store>
__device__ __forceinline__ void batched_for_b(int total, Load load, Store store) {
    using V = decltype(load(0));
    for (int e0 = threadIdx.x; e0 < total; e0 += B * NT) {
        V r[B];
#pragma unroll
        for (int it = 0; it < B; ++it) {
            const int e = e0 + NT * it;
            r[it] = load(EXACT || e < total ? e : total - 1);
        }
#pragma unroll
        for (int it = 0; it < B; ++it) {
            const int e = e0 + NT * it;
            if (EXACT || e < total) store(e, r[it]);
        }
    }
}
// batch size by the trip count (a short loop must not pad itself to eight clamped loads); total == 8 NT, the full
// tile of most kernels here, runs without clamps or guards
template <int NT, class Load, class Store>
__device__ __forceinline__ void batched_for(int total, Load load, Store store) {
    if (total == 8 * NT) batched_for_b<NT, 8, true>(total, load, store);
    else if (total <= 2 * NT) batched_for_b<NT, 2, false>(total, load, store);
    else if (total <= 4 * NT) batched_for_b<NT, 4, false>(total, load, store);
    else batched_for_b<NT, 8, false>(total, load, store);
}

// buffer-addressed 8-byte accesses: one VGPR byte offset per lane, the steps of an unrolled run in SGPRs / immediates (the
// flat form spends ~3 VALU + a carry-hazard nop on every 64-bit address); an access at or beyond `bytes` reads zero /
// is dropped, which makes the tail checks of a partial tile free
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
// Cache-policy operand of the buffer / global accesses (gfx950: 1 = sc0, 2 = nt, 16 = sc1).  Data a kernel streams through exactly once
// should neither evict the tables the resident workgroups re-read from L1 nor be kept in L2 / the Infinity Cache behind
// the store: loads sc1 (L1 bypassed), stores nt.  Measured on the headline stream (freq_kernels.hip): -7 % together.
// (-DYG_STREAM_LD= / -DYG_STREAM_ST= build the A/B variants of tools/ab_pkg.py.)
#ifndef YG_STREAM_LD
#define YG_STREAM_LD 16
#endif
#ifndef YG_STREAM_ST
#define YG_STREAM_ST 2
#endif
constexpr int kStreamLoad = YG_STREAM_LD, kStreamStore = YG_STREAM_ST;
__device__ __forceinline__ float2 buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {    // streamed data
    const v2u_t q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kStreamLoad);
    return make_float2(__uint_as_float(q.x), __uint_as_float(q.y));
}
// the same policies for accesses through plain pointers: a relaxed agent-scope atomic load IS `global_load ... sc1`,
// a non-temporal store `global_store ... nt` (4-, 8- and, stores only, 16-byte types)
template <class T>
__device__ __forceinline__ T ld_stream(const T *p) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte samples");
    if constexpr (kStreamLoad == 16) {
        T v;
        if constexpr (sizeof(T) == 4) {
            const unsigned q = __hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_memcpy(&v, &q, 4);
        } else {
            const unsigned long long q = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
            __builtin_memcpy(&v, &q, 8);
        }
        return v;
    } else {
        return *p;
    }
}
template <class T>
__device__ __forceinline__ void st_stream(T *p, T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16, "4-, 8- or 16-byte samples");
    if constexpr (kStreamStore == 2) {
        typedef float vec_t __attribute__((ext_vector_type(sizeof(T) / 4)));
        vec_t q;
        __builtin_memcpy(&q, &v, sizeof(T));
        __builtin_nontemporal_store(q, reinterpret_cast<vec_t *>(p));
    } else {
        *p = v;
    }
}
template <int AUX>
__device__ __forceinline__ float2 buf_ld_aux(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v2u_t q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    return make_float2(__uint_as_float(q.x), __uint_as_float(q.y));
}
template <int AUX>
__device__ __forceinline__ void buf_st_aux(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 v) {
    v2u_t q;
    q.x = __float_as_uint(v.x);
    q.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(q, r, voff, soff, AUX);
}
template <class T>
__device__ __forceinline__ T buf_ld_t(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (sizeof(T) == 4) {
        const unsigned q = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, kStreamLoad);
        T v;
        __builtin_memcpy(&v, &q, 4);
        return v;
    } else {
        static_assert(sizeof(T) == 8, "4- or 8-byte samples");
        const v2u_t q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, kStreamLoad);
        T v;
        __builtin_memcpy(&v, &q, 8);
        return v;
    }
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float2 v) {
    v2u_t q;
    q.x = __float_as_uint(v.x);
    q.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(q, r, voff, soff, kStreamStore);
}


}  // namespace yagi
