// micro-benchmark: issue rate of v_pk_fma_f32 with (a) all-VGPR operands, (b) an SGPR pair as src1 (how the direct-form
// FIR kernels feed their taps), (c) SGPR pair with the op_sel broadcast used for real taps.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ub/pkfma tools/ub/pkfma.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float s0, float s1, int iters) {
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.0f};
    v2f x = v2f{(float)threadIdx.x * 1e-3f, 0.5f};
    v2f hv = v2f{s0, s1};
    v2f hs = v2f{s0, s1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(hv));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(hs));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "s"(hs));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[i]) : "v"(x), "s"(hs));
            }
        }
    }
    v2f t = a[0];
    for (int i = 1; i < 8; ++i) t += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}

template <int MODE>
static void run(const char *name, float *out) {
    const int grid = 256 * 8, iters = 2000;     // 8 workgroups per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, 1.0001f, 0.9999f, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, 1.0001f, 0.9999f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 256 * iters * 64 * 4;      // 64 pk_fma per iteration, 4 flop each
    printf("%-34s %8.3f ms  %7.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("pk_fma vgpr,vgpr", out);
    run<1>("pk_fma vgpr,sgpr-pair", out);
    run<2>("pk_fma vgpr,sgpr lo-broadcast", out);
    run<3>("pk_fma vgpr hi-broadcast,sgpr-pair", out);
    run<0>("pk_fma vgpr,vgpr (again)", out);
    hipFree(out);
    return 0;
}
