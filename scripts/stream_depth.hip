// Per-CU streaming bandwidth against the number of 16-byte loads a thread keeps in flight, with ONE 256-thread workgroup per CU
// (150 KB of LDS claimed) on a subset of the CUs -- the regime of the solver kernel's triangular sweeps.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/build/stream_depth scripts/stream_depth.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d2 __attribute__((vector_size(16)));
template <int K>
__global__ void __launch_bounds__(256) stream_k(const d2* __restrict__ src, double* out, size_t per_wg_elems, int rounds)
{
    __shared__ double pad[150 * 128];                 // 150 KB: one workgroup per CU
    const d2* p = src + (size_t)blockIdx.x * per_wg_elems + threadIdx.x;
    d2 acc = {0.0, 0.0};
    for (int r = 0; r < rounds; ++r) {
        d2 v[K];
#pragma unroll
        for (int u = 0; u < K; ++u) v[u] = p[(size_t)(r * K + u) * 256];
#pragma unroll
        for (int u = 0; u < K; ++u) acc += v[u];
    }
    pad[threadIdx.x] = acc[0] + acc[1];
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = pad[0] + pad[255];
}
template <int K> void run(const d2* src, double* out, int wgs, size_t bytes_per_wg)
{
    const size_t elems = bytes_per_wg / 16;
    const int rounds = (int)(elems / 256 / K);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(stream_k<K>, dim3(wgs), dim3(256), 0, 0, src, out, elems, rounds);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double gb = (double)rounds * K * 256 * 16 * wgs / 1e9;
    printf("{\"wgs\": %d, \"loads_in_flight_per_thread\": %d, \"KB_in_flight_per_cu\": %.0f, \"GBps_total\": %.0f, \"GBps_per_cu\": %.1f}\n", wgs, K, K * 256 * 16 / 1024.0, gb / (ms * 1e-3), gb / (ms * 1e-3) / wgs);
}
int main()
{
    const size_t per = 64ull << 20;                  // 64 MB per workgroup: nothing is re-read
    for (int wgs : {32, 64, 128, 256}) {
        d2* src; double* out;
        if (hipMalloc((void**)&src, per * wgs) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMalloc((void**)&out, 8 * wgs);
        hipMemset(src, 0, per * wgs);
        run<4>(src, out, wgs, per); run<8>(src, out, wgs, per); run<16>(src, out, wgs, per); run<32>(src, out, wgs, per); run<64>(src, out, wgs, per);
        hipFree(src); hipFree(out);
    }
    return 0;
}
