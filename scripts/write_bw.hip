// write_bw.hip -- DIAGNOSTIC (not part of the library): what the memory system of one MI355X delivers to the access patterns of the
// saddle-point elimination, 512 workgroups of 256 threads (two per CU, as the solver kernel runs), every workgroup streaming through its
// own contiguous region of `kb` KB, several repetitions:
//   read16   16-byte loads, fully coalesced                      (the spike pass / the chains of a solve)
//   write8   8-byte stores, fully coalesced                      (a stream of doubles)
//   write16  16-byte stores, fully coalesced
//   writerec 8-byte stores in the elimination's pattern: per step and 16-lane group, lanes 0..4 put 20 doubles into a 160-byte record,
//            lanes 11..15 25 doubles into a 208-byte record of a second array (consecutive steps: consecutive records)
//   mix      writerec + read16 of an equally large region at the same time (half the workgroups each)
//   hipcc --offload-arch=gfx950 -O3 -o write_bw scripts/write_bw.hip ;  ./write_bw [kb per workgroup 2048] [reps 5]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d2 __attribute__((vector_size(16)));

__global__ void __launch_bounds__(256, 2) k_read16(const d2* a, size_t per, double* sink)
{
    const d2* p = a + (size_t)blockIdx.x * per;
    d2 acc = {0.0, 0.0};
    for (size_t i = threadIdx.x; i < per; i += 256 * 4) {
        d2 v0 = p[i], v1 = i + 256 < per ? p[i + 256] : acc, v2 = i + 512 < per ? p[i + 512] : acc, v3 = i + 768 < per ? p[i + 768] : acc;
        acc += v0 + v1 + v2 + v3;
    }
    if (acc[0] + acc[1] == 1.2345e-300) *sink = acc[0];
}
__global__ void __launch_bounds__(256, 2) k_write8(double* a, size_t per)
{
    double* p = a + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += 256) p[i] = (double)i;
}
__global__ void __launch_bounds__(256, 2) k_write16(d2* a, size_t per)
{
    d2* p = a + (size_t)blockIdx.x * per;
    const d2 v = {1.0, 2.0};
    for (size_t i = threadIdx.x; i < per; i += 256) p[i] = v;
}
// per: doubles of the region of one workgroup, split 20 : 26 between the two record arrays; 16 groups of 16 lanes, each with its own segment
__device__ __forceinline__ void writerec_body(double* a, size_t per)
{
    double* base = a + (size_t)blockIdx.x * per;
    const int g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const size_t steps = per / (16 * 46);
    double* AD = base + (size_t)g * steps * 20;
    double* AY = base + (size_t)16 * steps * 20 + (size_t)g * steps * 26;
    for (size_t k = 0; k < steps; ++k) {
        const double v = (double)k;
        if (cl < 5) {
            for (int r = cl; r < 5; ++r) AD[k * 20 + r * (r + 1) / 2 + cl] = v;
            AD[k * 20 + 15 + cl] = v;
        } else if (cl >= 11 && cl < 15) {
            for (int r = 0; r < 5; ++r) AY[k * 26 + r * 4 + (cl - 11)] = v;
        } else if (cl == 15) {
            for (int r = 0; r < 5; ++r) AY[k * 26 + 20 + r] = v;
        }
    }
}
__global__ void __launch_bounds__(256, 2) k_writerec(double* a, size_t per) { writerec_body(a, per); }
// the same stores, STEP-MAJOR layout: record slot = step * 16 + group -- the sixteen groups of a workgroup interleave into TWO write streams
// (all AD records of a step are 2560 contiguous bytes, all AY records 3328) instead of thirty-two
__global__ void __launch_bounds__(256, 2) k_writerec_sm(double* a, size_t per)
{
    double* base = a + (size_t)blockIdx.x * per;
    const int g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const size_t steps = per / (16 * 46);
    double* AD = base;
    double* AY = base + (size_t)16 * steps * 20;
    for (size_t k = 0; k < steps; ++k) {
        const double v = (double)k;
        const size_t slot = k * 16 + g;
        if (cl < 5) {
            for (int r = cl; r < 5; ++r) AD[slot * 20 + r * (r + 1) / 2 + cl] = v;
            AD[slot * 20 + 15 + cl] = v;
        } else if (cl >= 11 && cl < 15) {
            for (int r = 0; r < 5; ++r) AY[slot * 26 + r * 4 + (cl - 11)] = v;
        } else if (cl == 15) {
            for (int r = 0; r < 5; ++r) AY[slot * 26 + 20 + r] = v;
        }
    }
}
// per-group streams, scattered by lane as in writerec, but every lane's values are CONTIGUOUS in the record (column-major-by-lane record
// layout: lane c < 5 owns 6 - c doubles at {0, 6, 11, 15, 18}[c], lanes 11..15 five doubles each) and go out as 16-byte + 8-byte pieces:
// at most three store instructions per step instead of six
typedef double d2u __attribute__((vector_size(16), aligned(8)));
template <bool SM> __device__ __forceinline__ void writerec_lc_body(double* a, size_t per)
{
    double* base = a + (size_t)blockIdx.x * per;
    const int g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const size_t steps = per / (16 * 46);
    double* AD = SM ? base : base + (size_t)g * steps * 20;
    double* AY = base + (size_t)16 * steps * 20 + (SM ? 0 : (size_t)g * steps * 26);
    const int cb = cl == 0 ? 0 : cl == 1 ? 6 : cl == 2 ? 11 : cl == 3 ? 15 : 18;
    const int cnt = cl < 5 ? 6 - cl : 5;
    for (size_t k = 0; k < steps; ++k) {
        const size_t slot = SM ? k * 16 + g : k;
        double* o = cl < 5 ? AD + slot * 20 + cb : AY + slot * 26 + (cl - 11) * 5;
        if (cl >= 5 && cl < 11) continue;
        const d2u v = {(double)k, 1.0};
        if (cnt >= 2) *(d2u*)o = v;
        if (cnt >= 4) *(d2u*)(o + 2) = v;
        if (cnt >= 6) *(d2u*)(o + 4) = v;
        if (cnt & 1) o[cnt - 1] = (double)k;
    }
}
__global__ void __launch_bounds__(256, 2) k_writerec_lc(double* a, size_t per) { writerec_lc_body<false>(a, per); }
__global__ void __launch_bounds__(256, 2) k_writerec_lc_sm(double* a, size_t per) { writerec_lc_body<true>(a, per); }
// writerec with the records of TWO consecutive steps interleaved entry by entry (entry e of steps 2j, 2j + 1 adjacent): every lane keeps the
// values of the even step in registers and writes 16 bytes every second step -- half the store instructions and half the write requests
// for the same bytes, no LDS staging.  ROWS: the y lane's five values are the fifth column of the spike record (five lanes x 16 bytes per
// instruction) instead of five separate 16-byte stores
template <bool ROWS> __device__ __forceinline__ void writerec_pair_body(double* a, size_t per)
{
    double* base = a + (size_t)blockIdx.x * per;
    const int g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const size_t steps = per / (16 * 46);
    double* AD = base + (size_t)g * steps * 20;
    double* AY = base + (size_t)16 * steps * 20 + (size_t)g * steps * 26;
    for (size_t k = 0; k + 1 < steps; k += 2) {
        const d2u v = {(double)k, 1.0};
        const size_t j = k >> 1;
        if (cl < 5) {
            for (int r = cl; r < 5; ++r) *(d2u*)(AD + j * 40 + (r * (r + 1) / 2 + cl) * 2) = v;
            *(d2u*)(AD + j * 40 + (15 + cl) * 2) = v;
        } else if (ROWS && cl >= 11) {
            for (int r = 0; r < 5; ++r) *(d2u*)(AY + j * 52 + (r * 5 + (cl - 11)) * 2) = v;
        } else if (!ROWS && cl >= 11 && cl < 15) {
            for (int r = 0; r < 5; ++r) *(d2u*)(AY + j * 52 + (r * 4 + (cl - 11)) * 2) = v;
        } else if (!ROWS && cl == 15) {
            for (int r = 0; r < 5; ++r) *(d2u*)(AY + j * 52 + (20 + r) * 2) = v;
        }
    }
}
__global__ void __launch_bounds__(256, 2) k_writerec_pair(double* a, size_t per) { writerec_pair_body<false>(a, per); }
__global__ void __launch_bounds__(256, 2) k_writerec_pair_rows(double* a, size_t per) { writerec_pair_body<true>(a, per); }
// the same with FOUR steps interleaved (two 16-byte stores per entry and lane every fourth step, 32 bytes per lane contiguous)
__global__ void __launch_bounds__(256, 2) k_writerec_quad_rows(double* a, size_t per)
{
    double* base = a + (size_t)blockIdx.x * per;
    const int g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const size_t steps = per / (16 * 46);
    double* AD = base + (size_t)g * steps * 20;
    double* AY = base + (size_t)16 * steps * 20 + (size_t)g * steps * 26;
    for (size_t k = 0; k + 3 < steps; k += 4) {
        const d2u v = {(double)k, 1.0};
        const size_t j = k >> 2;
        if (cl < 5) {
            for (int r = cl; r < 5; ++r) { double* o = AD + j * 80 + (r * (r + 1) / 2 + cl) * 4; *(d2u*)o = v; *(d2u*)(o + 2) = v; }
            double* o = AD + j * 80 + (15 + cl) * 4; *(d2u*)o = v; *(d2u*)(o + 2) = v;
        } else if (cl >= 11) {
            for (int r = 0; r < 5; ++r) { double* o = AY + j * 104 + (r * 5 + (cl - 11)) * 4; *(d2u*)o = v; *(d2u*)(o + 2) = v; }
        }
    }
}
// per-group streams as in writerec, but every store instruction writes contiguous 16-byte items (what staging the records of two steps in
// LDS produces: 46 items per pair of steps and group)
__global__ void __launch_bounds__(256, 2) k_writerec_staged(double* a, size_t per)
{
    double* base = a + (size_t)blockIdx.x * per;
    const int g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const size_t steps = per / (16 * 46);
    d2* S = (d2*)(base + (size_t)g * steps * 46);
    const d2 v = {1.0, 2.0};
    for (size_t k = 0; k + 1 < steps; k += 2)
        for (int j = 0; j < 3; ++j) {
            const int idx = cl + 16 * j;
            if (idx < 46) S[k * 23 + idx] = v;
        }
}
// odd workgroups write records, even ones read 16-byte items: both kinds resident on every CU at the same time
__global__ void __launch_bounds__(256, 2) k_mix(double* a, size_t per, double* sink)
{
    if (blockIdx.x & 1) { writerec_body(a, per); return; }
    const d2* p = (const d2*)(a + (size_t)blockIdx.x * per);
    d2 acc = {0.0, 0.0};
    for (size_t i = threadIdx.x; i < per / 2; i += 256) acc += p[i];
    if (acc[0] + acc[1] == 1.2345e-300) *sink = acc[0];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const size_t kb = argc > 1 ? (size_t)atoi(argv[1]) : 2048;
    const int reps = argc > 2 ? atoi(argv[2]) : 5, wgs = 512;
    const size_t per_bytes = kb * 1024, total = per_bytes * wgs;
    double *a = nullptr, *sink = nullptr;
    CK(hipMalloc((void**)&a, total));
    CK(hipMalloc((void**)&sink, 64));
    CK(hipMemset(a, 0, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[] = {"read16", "write8", "write16", "writerec", "mix_writerec_read16", "writerec_stepmajor", "writerec_staged", "writerec_lanecontig", "writerec_lanecontig_stepmajor", "writerec_pair", "writerec_pair_rows", "writerec_quad_rows"};
    for (int t = 0; t < 12; ++t) {
        float best = 1e30f;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(e0, 0));
            if (t == 0) hipLaunchKernelGGL(k_read16, dim3(wgs), dim3(256), 0, 0, (const d2*)a, per_bytes / 16, sink);
            if (t == 1) hipLaunchKernelGGL(k_write8, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            if (t == 2) hipLaunchKernelGGL(k_write16, dim3(wgs), dim3(256), 0, 0, (d2*)a, per_bytes / 16);
            if (t == 3) hipLaunchKernelGGL(k_writerec, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            if (t == 4) hipLaunchKernelGGL(k_mix, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8, sink);
            if (t == 5) hipLaunchKernelGGL(k_writerec_sm, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            if (t == 6) hipLaunchKernelGGL(k_writerec_staged, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            if (t == 7) hipLaunchKernelGGL(k_writerec_lc, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            if (t == 8) hipLaunchKernelGGL(k_writerec_lc_sm, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            if (t == 9) hipLaunchKernelGGL(k_writerec_pair, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            if (t == 10) hipLaunchKernelGGL(k_writerec_pair_rows, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            if (t == 11) hipLaunchKernelGGL(k_writerec_quad_rows, dim3(wgs), dim3(256), 0, 0, a, per_bytes / 8);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
        }
        const double rec = (double)(per_bytes / 8 / (16 * 46)) * 16 * 45 * 8;          // bytes one workgroup writes in the record pattern
        const double moved = (t == 3 || t == 5 || t == 7 || t >= 8) ? rec * wgs : t == 4 ? (rec + (double)per_bytes) * (wgs / 2)
                             : t == 6 ? (double)(per_bytes / 8 / (16 * 46) / 2) * 2 * 16 * 46 * 8 * wgs : (double)total;
        printf("{\"pattern\": \"%s\", \"workgroups\": %d, \"KB_per_workgroup\": %zu, \"best_ms\": %.3f, \"GBps\": %.0f}\n", names[t], wgs, kb, best,
               moved / (best * 1e-3) / 1e9);
    }
    return 0;
}
