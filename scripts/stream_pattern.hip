// The loader waves' access pattern of the triangular sweeps in isolation: 192 threads of a 256-thread workgroup
// (one per CU, 150 KB of LDS claimed) fetch a 74 KB "chunk" per step into registers, write it to LDS one step later, one workgroup
// barrier per step.  Pattern 0: the whole chunk is requested in one burst per step (what the solver does); pattern 1: two half
// chunks, half a step apart (a second barrier in the middle); pattern 2: four quarters.  Same registers, same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/stream_pattern scripts/stream_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d2 __attribute__((vector_size(16)));
#define ITEMS 24                       /* 24 x 16 B x 192 threads = 73.7 KB per step */
template <int PARTS>
__global__ void __launch_bounds__(256) pattern_k(const d2* __restrict__ src, double* out, size_t per_wg_elems, int steps)
{
    __shared__ d2 lds[ITEMS * 192 + 64];
    __shared__ double pad[140 * 128 - 2 * (ITEMS * 192 + 64)];       // about 150 KB in total: one workgroup per CU
    const int tid = threadIdx.x, lt = tid - 64;
    const d2* p = src + (size_t)blockIdx.x * per_wg_elems + (lt >= 0 ? lt : 0);
    d2 regs[ITEMS];
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) regs[u] = (d2){0.0, 0.0};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
            if (lt >= 0) {
#pragma unroll
                for (int u = part * (ITEMS / PARTS); u < (part + 1) * (ITEMS / PARTS); ++u) lds[u * 192 + lt] = regs[u];     // commit (loaded a step ago)
#pragma unroll
                for (int u = part * (ITEMS / PARTS); u < (part + 1) * (ITEMS / PARTS); ++u) regs[u] = p[((size_t)s * ITEMS + u) * 192];
            }
            __syncthreads();
        }
    }
    if (lt >= 0) { double t = 0.0; for (int u = 0; u < ITEMS; ++u) t += regs[u][0]; pad[tid] = t + lds[lt][0]; } else pad[tid] = acc;
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = pad[0] + pad[255];
}
template <int PARTS> void run(const d2* src, double* out, int wgs, size_t bytes_per_wg)
{
    const size_t elems = bytes_per_wg / 16;
    const int steps = (int)(elems / (ITEMS * 192));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(pattern_k<PARTS>, dim3(wgs), dim3(256), 0, 0, src, out, elems, steps);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
    }
    const double gb = (double)steps * ITEMS * 192 * 16 * wgs / 1e9;
    printf("{\"wgs\": %d, \"parts_per_step\": %d, \"us_per_step\": %.2f, \"GBps_per_cu\": %.1f}\n", wgs, PARTS, ms * 1e3 / steps, gb / (ms * 1e-3) / wgs);
}
int main()
{
    const size_t per = 48ull << 20;
    for (int wgs : {64, 128, 256}) {
        d2* src; double* out;
        if (hipMalloc((void**)&src, per * wgs) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMalloc((void**)&out, 8 * wgs); hipMemset(src, 0, per * wgs);
        run<1>(src, out, wgs, per); run<2>(src, out, wgs, per); run<4>(src, out, wgs, per);
        hipFree(src); hipFree(out);
    }
    return 0;
}
