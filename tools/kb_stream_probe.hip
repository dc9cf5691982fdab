// kb_stream_probe.hip -- what a kernel that does nothing but move float2 frames reaches on this device for a given
// read : write mix (diagnostic, not part of the library).  Build here, run on the GPU box (the binary travels with gpurun):
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result tools/kb_stream_probe.hip -o tools/kb_stream_probe
//   gpurun -- 'tools/kb_stream_probe > gpurun_out/stream_probe.txt'        (profiles/r03_stream_probe.txt)
// One 256-lane workgroup per 32 KiB unit, like the path's frame kernels: R loads and W stores of 2 KiB rows per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v2f_t __attribute__((ext_vector_type(2)));

template <int R, int W, bool NT>
__global__ void __launch_bounds__(256) probe(const float2 *__restrict__ x, float2 *__restrict__ y, float k) {
    const size_t u = blockIdx.x;
    const float2 *src = x + u * (size_t)(256 * R) + threadIdx.x;
    float2 *dst = y + u * (size_t)(256 * W) + threadIdx.x;
    float2 v[R > 0 ? R : 1];
    if (R > 0) {
#pragma unroll
        for (int a = 0; a < R; ++a) { const v2f_t q = __builtin_nontemporal_load(reinterpret_cast<const v2f_t *>(src + 256 * a)); v[a] = make_float2(q.x, q.y); }
    } else {
        v[0] = make_float2(k, k);
    }
    if (W > 0) {
#pragma unroll
        for (int a = 0; a < W; ++a) {
            float2 o = make_float2(0.f, 0.f);                  // every load reaches a store
            constexpr int RR = R > 0 ? R : 1;
#pragma unroll
            for (int j = a % RR; j < RR; j += (W < RR ? W : RR)) {
                o.x += v[j].x * k;
                o.y += v[j].y;
            }
            if (NT) __builtin_nontemporal_store(v2f_t{o.x, o.y}, reinterpret_cast<v2f_t *>(dst + 256 * a));
            else dst[256 * a] = o;
        }
    } else {                                   // read only: a store that never happens keeps the loads alive
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < R; ++a) s += v[a].x + v[a].y;
        if (s == 12345.678f) y[threadIdx.x] = make_float2(s, s);
    }
}

template <int R, int W, bool NT>
static void run(const char *name, const float2 *x, float2 *y, size_t units) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) probe<R, W, NT><<<(unsigned)units, 256>>>(x, y, 0.5f);
    const int reps = 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) probe<R, W, NT><<<(unsigned)units, 256>>>(x, y, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)units * 256.0 * 8.0 * (R + W);
    printf("%-44s %8.4f ms  %6.2f TB/s = %.3f of 8 TB/s\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
}

int main() {
    const size_t units = 1u << 16;                       // x 32 KiB x 16 rows: 2 GiB in, 2 GiB out at most
    float2 *x, *y;
    if (hipMalloc(&x, units * 256 * 16 * sizeof(float2)) != hipSuccess) return 1;
    if (hipMalloc(&y, units * 256 * 16 * sizeof(float2)) != hipSuccess) return 1;
    hipMemset(x, 0x3c, units * 256 * 16 * sizeof(float2));
    hipMemset(y, 0, units * 256 * 16 * sizeof(float2));
    run<16, 16, false>("read 1 : write 1", x, y, units);
    run<16, 16, true>("read 1 : write 1, nt stores", x, y, units);
    run<8, 16, false>("read 1 : write 2 (firpfbch2 analyzer)", x, y, units);
    run<8, 16, true>("read 1 : write 2, nt stores", x, y, units);
    run<16, 8, false>("read 2 : write 1 (resamp2 decimator)", x, y, units);
    run<16, 8, true>("read 2 : write 1, nt stores", x, y, units);
    run<16, 2, false>("read 8 : write 1 (msresamp2 / 8)", x, y, units);
    run<16, 2, true>("read 8 : write 1, nt stores", x, y, units);
    run<16, 0, false>("read only", x, y, units);
    run<0, 16, false>("write only", x, y, units);
    run<0, 16, true>("write only, nt stores", x, y, units);
    return 0;
}
