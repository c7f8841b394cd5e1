// factor_bench.hip -- DIAGNOSTIC (not part of the library): the bordered-band factorisation of mcq_kernels.hip in isolation, one
// workgroup per problem, `reps` factorisations of a synthetic SPD bordered band per workgroup, with compile-time ablation switches
// (-DMCQ_ABL=mask, see factor_t) that REMOVE parts of a step -- the results are then garbage, only the time is looked at: what a
// part costs ON THE CRITICAL PATH of a step is the time that disappears with it.  Same grid shape as the solver (1024 workgroups on
// 256 CUs), so the memory system sees the factorisation's own traffic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -enable-ipra=0 -DMCQ_CORE_BAND [-DMCQ_ABL=..] -o fb scripts/factor_bench.hip
// (the bordered-band core of rounds 1-3: since the saddle-point core it only serves the shortest-path objective; scripts/kkt_check.hip is
//  the diagnostic of the core the headline path runs on)
//   ./fb [batch 1024] [n 2000] [reps 12] [with_fwd 1]
#include "../global_racetrajectory_optimization_amd/csrc/mcq_kernels.hip"      // compiled with -DMCQ_CORE_BAND: the bordered-band core lives in namespace mcq_band
using namespace mcq_band;

#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

__global__ void __launch_bounds__(MCQ_NT) fb_kernel(McqBatch B, int reps, int with_fwd, long long* out)
{
    int n;
    double kb, wv;
    SolveCtx c;
    c.w = mcq_work(B, blockIdx.x, n, kb, wv);
    c.nm = B.nmax;
    c.d = mcq_dims(n, B.band_e);
    for (int q = 0; q < 8; ++q) c.tk[q] = 0;
    c.last_step = 0.0;
    c.refine_rounds = c.second_attempt = 0;
    c.direct = 0;
    gdouble* SIG = VEC(c.w, c.nm, V_SIG);
    gdouble* RHS = VEC(c.w, c.nm, V_RHS);
    if (MCQ_WORKER_TIMERS && threadIdx.x == 64 * (MCQ_WORKER_TIMERS & 3)) for (int q = 0; q < 8; ++q) ((long long*)c.w.Z)[q] = 0;
    __syncthreads();
    const long long t0 = (long long)clock64();
    int fs = 0;
    for (int r = 0; r < reps; ++r) {
        for (int i = threadIdx.x; i < n; i += MCQ_NT) RHS[i] = 1.0;
        __syncthreads();
        fs |= factor(c, c.w.H, SIG, nullptr, with_fwd ? RHS : nullptr);
    }
    const long long t1 = (long long)clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = fs; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int batch = argc > 1 ? atoi(argv[1]) : 1024, n = argc > 2 ? atoi(argv[2]) : 2000, reps = argc > 3 ? atoi(argv[3]) : 12;
    const int with_fwd = argc > 4 ? atoi(argv[4]) : 1;
    const McqDims d = mcq_dims(n, 32);
    const size_t elems = (size_t)batch * n;
    std::vector<double> Hh((size_t)n * MCQ_HLD, 0.0);
    for (int i = 0; i < d.ni; ++i) {
        Hh[MCQ_HBAND(i, 0)] = 4.0;
        for (int k = 1; k <= d.b && i + k < d.ni; ++k) Hh[MCQ_HBAND(i, k)] = 0.5 / ((1.0 + k) * (1.0 + k));
        if (i < d.p || i >= d.ni - d.p)
            for (int jj = 0; jj < d.p; ++jj) Hh[(size_t)i * MCQ_HLD + MCQ_HBO + jj] = 0.002 / (1.0 + ((i + jj) % 7));
    }
    for (int j = 0; j < d.p; ++j)
        for (int jj = 0; jj < d.p; ++jj) Hh[(size_t)(d.ni + j) * MCQ_HLD + MCQ_HBO + jj] = j == jj ? 4.0 : 0.01 / (1.0 + abs(j - jj));
    double *H, *L, *vec, *Z;
    signed char* state;
    int* status;
    long long* out;
    CK(hipMalloc((void**)&H, elems * MCQ_HLD * sizeof(double)));
    CK(hipMalloc((void**)&L, elems * MCQ_LLD * sizeof(double)));
    CK(hipMalloc((void**)&vec, elems * MCQ_NVEC * sizeof(double)));
    CK(hipMalloc((void**)&Z, (elems + (size_t)batch * MCQ_KMAX * MCQ_KMAX) * sizeof(double)));
    CK(hipMalloc((void**)&state, elems));
    CK(hipMalloc((void**)&status, batch * sizeof(int)));
    CK(hipMalloc((void**)&out, batch * 2 * sizeof(long long)));
    CK(hipMemset(state, 0, elems));
    CK(hipMemset(status, 0, batch * sizeof(int)));
    CK(hipMemset(L, 0, elems * MCQ_LLD * sizeof(double)));
    for (int b = 0; b < batch; ++b) CK(hipMemcpy(H + (size_t)b * n * MCQ_HLD, Hh.data(), Hh.size() * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<double> v((size_t)MCQ_NVEC * n, 1.0);
        for (int b = 0; b < batch; ++b) CK(hipMemcpy(vec + (size_t)b * n * MCQ_NVEC, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch; B.n = n; B.nmax = n;
    B.ref = H; B.nv = nullptr; B.sc = nullptr;         // never dereferenced here
    B.Eb = B.Et = B.Db = L;                             // never dereferenced here
    B.H = H; B.L = L; B.vec = vec; B.Z = Z; B.state = state; B.status = status;
    B.band_e = 32;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fb_kernel, dim3(batch), dim3(MCQ_NT), 0, 0, B, reps, with_fwd, out);
        CK(hipGetLastError());
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<long long> o((size_t)batch * 2);
    CK(hipMemcpy(o.data(), out, o.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double cyc = 0.0;
    int bad = 0;
    for (int b = 0; b < batch; ++b) { cyc += (double)o[2 * b]; bad += o[2 * b + 1] != 0; }
    cyc /= batch;
    const int nblk = (d.ni + 15) / 16;
    if (MCQ_WORKER_TIMERS) {
        // per-step cycles of the sampled wave (problem 0): the accumulators factor_t leaves in the curvature-row scratch
        long long wt[8];
        CK(hipMemcpy(wt, Z, sizeof(wt), hipMemcpyDeviceToHost));
        printf("worker %d cycles/step:", (int)MCQ_WORKER_TIMERS);
        for (int q = 0; q < 8; ++q) printf(" [%d] %.0f", q, (double)wt[q] / reps / nblk);
        printf("\n");
    }
    printf("{\"abl\": %d, \"batch\": %d, \"n\": %d, \"reps\": %d, \"with_fwd\": %d, \"kernel_ms\": %.3f, \"us_per_factorisation_per_workgroup\": %.2f, "
           "\"cycles_per_factorisation\": %.0f, \"cycles_per_step\": %.0f, \"not_pd\": %d}\n",
           (int)MCQ_ABL, batch, n, reps, with_fwd, best, 1e3 * best / reps / ((batch + 255) / 256), cyc / reps, cyc / reps / nblk, bad);
    return 0;
}
