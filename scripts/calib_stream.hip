// Calibration of the HBM traffic counters (FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ*) on streams of KNOWN size and of the access
// shapes mcq_solve_kernel uses (MI355X_MICROARCH.md, section HBM: "calibrate on a known byte count in your own access pattern"):
//   read8   every lane loads 8 bytes  (global_load_dwordx2: the band products, the H-row fetch of the factorisation)
//   read16  every lane loads 16 bytes (global_load_dwordx4: the L / W rows of the triangular sweeps)
//   write8  every lane stores 8 bytes
// over a buffer of BYTES bytes (default 2 GiB: far beyond the 256 MiB Infinity Cache), each exactly once.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/build/calib_stream scripts/calib_stream.hip      (scripts/profile_round.sh)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d2 __attribute__((vector_size(16)));

__global__ void calib_read8(const double* src, double* sink, size_t n)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 1.2345e300) sink[0] = acc;
}
__global__ void calib_read16(const d2* src, double* sink, size_t n2)
{
    d2 acc = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc[0] + acc[1] == 1.2345e300) sink[0] = acc[0];
}
__global__ void calib_write8(double* dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)i;
}

int main(int argc, char** argv)
{
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : ((size_t)2 << 30);
    const size_t n = bytes / 8;
    double *a = nullptr, *sink = nullptr;
    if (hipMalloc((void**)&a, bytes) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, (const double*)a, sink, n);
        hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const d2*)a, sink, n / 2);
        hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, a, n);
        hipDeviceSynchronize();
    }
    printf("{\"bytes_per_kernel\": %zu}\n", bytes);
    hipFree(a);
    hipFree(sink);
    return 0;
}
