// micro-benchmark: does f32 MFMA (v_mfma_f32_16x16x4_f32) co-execute with packed f32 VALU
// (v_pk_fma_f32) when the two run in different waves of the same SIMD?  Register-only loops.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_coexec.hip -o /tmp/ubench_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode bit0: waves with (wave&1)==0 ... roles: role = (mode==1) all MFMA, (mode==2) all VALU,
// (mode==3) waves 0..3 MFMA, waves 4..7 VALU (8-wave WG), (mode==4) all VALU plain v_fma
template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    bool do_mfma = (MODE == 1) || (MODE == 3 && wave < 4);
    bool do_valu = (MODE == 2) || (MODE == 3 && wave >= 4);
    float r = 0.f;
    if (do_mfma) {
        f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
        float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
            }
        }
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
    if (do_valu) {
        f32x2 acc[16];
        f32x2 xv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc[j] = f32x2{0.f, 0.f}; xv[j] = f32x2{seed + j, seed - j + threadIdx.x}; }
        float hk = seed * 0.25f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    acc[j] = __builtin_elementwise_fma(xv[(j + u) & 15], f32x2{hk, hk}, acc[j]);
                }
                hk += 1e-9f;
            }
        }
        for (int j = 0; j < 16; ++j) r += acc[j][0] + acc[j][1];
    }
    if (MODE == 4) {
        float acc[32], xv[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) { acc[j] = 0.f; xv[j] = seed + j + threadIdx.x; }
        float hk = seed * 0.25f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int j = 0; j < 32; ++j) acc[j] = __builtin_fmaf(xv[(j + u) & 31], hk, acc[j]);
                hk += 1e-9f;
            }
        }
        for (int j = 0; j < 32; ++j) r += acc[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
double run(float *d, int grid, int block, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<grid, block>>>(d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    float *d; hipMalloc(&d, 1 << 26);
    const int iters = 2000;
    // per wave per iter: MFMA 32 x 2048 flop ; VALU 256 pk_fma x 256 flop ; plain 512 fma x 128 flop
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        int grid = 256 * wg_per_cu;
        double t1 = run<1>(d, grid, 256, iters);   // 4 MFMA waves / WG
        double t2 = run<2>(d, grid, 256, iters);   // 4 VALU(pk) waves / WG
        double t4 = run<4>(d, grid, 256, iters);   // 4 VALU(plain fma) waves / WG
        double t3 = run<3>(d, grid, 512, iters);   // 4 MFMA + 4 VALU(pk) waves / WG
        double f_m = (double)grid * 4 * iters * 32 * 2048.0, f_v = (double)grid * 4 * iters * 256 * 256.0;
        double f_p = (double)grid * 4 * iters * 512 * 128.0;
        printf("WG/CU=%d  mfma-only %.3f ms %.1f TF | pk-valu-only %.3f ms %.1f TF | plain-fma-only %.3f ms %.1f TF | both %.3f ms %.1f TF (sum-of-alone %.3f)\n",
               wg_per_cu, t1, f_m / t1 / 1e9, t2, f_v / t2 / 1e9, t4, f_p / t4 / 1e9, t3, (f_m + f_v) / t3 / 1e9, t1 + t2);
    }
    return 0;
}
