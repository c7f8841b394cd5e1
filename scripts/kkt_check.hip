// kkt_check.hip -- DIAGNOSTIC (not part of the library): the saddle-point elimination of mcq_kkt.inc in isolation.  One workgroup per problem
// copy, `reps` factorisations + solves of the same system; every solution is compared with the first one (determinism: races show up as
// differences between repetitions) and with a dense LU of the same saddle-point system on the host (correctness).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -enable-ipra=0 --gpu-max-threads-per-block=512 [-DKKT_TIMERS=1] [-DKC_F32=1] -o kc scripts/kkt_check.hip ;  ./kc [n 333] [reps 50] [batch 4] [sigma exponent range 12] [host reference 1] [fused 0] [pinned fraction 0]
// (also builds against tests/emu: g++ -O2 -std=c++17 -x c++ -I tests/emu/include scripts/kkt_check.hip)
#include "../global_racetrajectory_optimization_amd/csrc/mcq_kernels.hip"
#ifndef KC_F32
#define KC_F32 0      /* 1: the records of the elimination stored as floats (KRec<true>): timing only -- the comparison with the dense LU then shows ~1e-7 */
#endif

#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <cmath>

__global__ void __launch_bounds__(MCQ_NT, 2) kc_kernel(McqBatch B, int reps, const double* rhs0, double* out, int* fsout, int fused, int masked, int nosig)
{
    int n;
    double kb, wv;
    ctx_set_problem(B, blockIdx.x, n, kb, wv);
    const LCtx& c = G_CTX;
    if (threadIdx.x == 0) {
        for (int q = 0; q < 8; ++q) g_ctx.tk[q] = 0;
        g_ctx.last_step = 0.0;
        g_ctx.refine_rounds = g_ctx.second_attempt = g_ctx.f32_count = 0;
        g_ctx.direct = 0;
        g_ctx.kkt_w = nullptr;
        g_ctx.kkt_f32 = 0;
    }
    __syncthreads();
    gdouble* SIG = VEC(c.w, c.nm, V_SIG);
    gdouble* RHS = VEC(c.w, c.nm, V_RHS);
    int fs = 0;
    long long tf = 0, ts = 0;
    for (int r = 0; r < reps; ++r) {
        for (int i = threadIdx.x; i < n; i += MCQ_NT) RHS[i] = rhs0[i];
        __syncthreads();
        const long long t0 = (long long)wall_clock64();
        fs |= factor_kkt<KC_F32 != 0>(c, nosig ? nullptr : SIG, masked ? c.w.state : nullptr, nullptr, fused ? RHS : nullptr);
        const long long t1 = (long long)wall_clock64();
        solve_kkt<KC_F32 != 0>(c, RHS, fused != 0);
        const long long t2 = (long long)wall_clock64();
        tf += t1 - t0; ts += t2 - t1;
        for (int i = threadIdx.x; i < n; i += MCQ_NT) out[((size_t)blockIdx.x * reps + r) * n + i] = RHS[i];
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0 && KKT_TIMERS) printf("   phases (us per call): factor fwd %.1f spikes %.1f separators %.1f | solve fwd %.1f bwd %.1f separators %.1f correction %.1f\n", c.tk[0] / 100.0 / reps, c.tk[1] / 100.0 / reps, c.tk[2] / 100.0 / reps, c.tk[3] / 100.0 / reps, c.tk[4] / 100.0 / reps, c.tk[5] / 100.0 / reps, c.tk[6] / 100.0 / reps);
    if (threadIdx.x == 0) { fsout[3 * blockIdx.x] = fs; fsout[3 * blockIdx.x + 1] = (int)(tf / reps); fsout[3 * blockIdx.x + 2] = (int)(ts / reps); }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double urand(unsigned long long& s)
{
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)(s >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 333, reps = argc > 2 ? atoi(argv[2]) : 50, batch = argc > 3 ? atoi(argv[3]) : 4;
    const double srange = argc > 4 ? atof(argv[4]) : 12.0;
    const int with_ref = argc > 5 ? atoi(argv[5]) : 1;
    const int fused = argc > 6 ? atoi(argv[6]) : 0;
    const double pin_frac = argc > 7 ? atof(argv[7]) : 0.0;        // fraction of pinned waypoints (rows / columns of the reduced system replaced by identity)
    const size_t elems = (size_t)batch * n;
    unsigned long long seed = 12345;
    // geometry of an oval: reference derivatives and unit normals; spline scalings near one
    std::vector<double> vec((size_t)MCQ_NVEC * n, 0.0), rhs(n);
    for (int i = 0; i < n; ++i) {
        const double th = 2.0 * M_PI * i / n;
        const double xp = -2.0 * sin(th) * 3.0, yp = cos(th) * 3.0;
        const double nrm = sqrt(xp * xp + yp * yp);
        vec[(size_t)V_XP * n + i] = xp;
        vec[(size_t)V_YP * n + i] = yp;
        vec[(size_t)V_CP * n + i] = 1.0 / (nrm * nrm * nrm);
        vec[(size_t)V_NX * n + i] = yp / nrm;
        vec[(size_t)V_NY * n + i] = -xp / nrm;
        vec[(size_t)V_SC * n + i] = 1.0 + 0.05 * sin(3.0 * th);
        vec[(size_t)V_SIG * n + i] = srange < -90.0 ? 0.0 : pow(10.0, -6.0 + (srange + 6.0) * urand(seed));      // sigma range < -90: no diagonal at all (the active-set phase's systems)
        rhs[i] = 2.0 * urand(seed) - 1.0;
    }
    double *L, *vecd, *rhsd, *outd;
    signed char* state;
    int *status, *fsd;
    CK(hipMalloc((void**)&L, elems * MCQ_LLD * sizeof(double)));
    CK(hipMalloc((void**)&vecd, elems * MCQ_NVEC * sizeof(double)));
    CK(hipMalloc((void**)&rhsd, n * sizeof(double)));
    CK(hipMalloc((void**)&outd, elems * reps * sizeof(double)));
    CK(hipMalloc((void**)&state, elems));
    CK(hipMalloc((void**)&status, batch * sizeof(int)));
    CK(hipMalloc((void**)&fsd, 3 * batch * sizeof(int)));
    CK(hipMemset(state, 0, elems));
    std::vector<signed char> pin(n, 0);
    if (pin_frac > 0.0) {
        for (int i = 0; i < n; ++i) pin[i] = urand(seed) < pin_frac ? 1 : 0;
        for (int b = 0; b < batch; ++b) CK(hipMemcpy(state + (size_t)b * n, pin.data(), n, hipMemcpyHostToDevice));
    }
    CK(hipMemset(status, 0, batch * sizeof(int)));
    CK(hipMemset(L, 0xff, elems * MCQ_LLD * sizeof(double)));
    for (int b = 0; b < batch; ++b) CK(hipMemcpy(vecd + (size_t)b * n * MCQ_NVEC, vec.data(), vec.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemcpy(rhsd, rhs.data(), n * sizeof(double), hipMemcpyHostToDevice));
    McqBatch B;
    memset(&B, 0, sizeof(B));
    B.batch = batch; B.n = n; B.nmax = n;
    B.L = L; B.vec = vecd; B.state = state; B.status = status;
    B.Z = L;      // unused by the saddle-point path
    B.ref = L; B.kappa_bound = 1.0; B.w_veh = 0.0;
    hipLaunchKernelGGL(kc_kernel, dim3(batch), dim3(MCQ_NT), 0, 0, B, reps, rhsd, outd, fsd, fused, pin_frac > 0.0 ? 1 : 0, srange < -90.0 ? 1 : 0);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<double> out(elems * reps);
    std::vector<int> fs(3 * batch);
    CK(hipMemcpy(out.data(), outd, out.size() * sizeof(double), hipMemcpyDeviceToHost));
    CK(hipMemcpy(fs.data(), fsd, 3 * batch * sizeof(int), hipMemcpyDeviceToHost));
    // determinism
    int ndiff = 0, nnan = 0;
    double dmax = 0.0;
    for (size_t q = 0; q < (size_t)batch * reps; ++q)
        for (int i = 0; i < n; ++i) {
            const double a = out[q * n + i], b0 = out[i];
            if (!(a == a)) { ++nnan; continue; }
            if (a != b0) { ++ndiff; dmax = fmax(dmax, fabs(a - b0)); }
        }
    { double af = 0, as_ = 0; for (int b = 0; b < batch; ++b) { af += fs[3 * b + 1]; as_ += fs[3 * b + 2]; }
      printf("   per workgroup: factorisation %.1f us, solve %.1f us (100 MHz wall clock, mean over %d workgroups)\n", af / batch / 100.0, as_ / batch / 100.0, batch); }
    printf("n %d reps %d batch %d: factor status %d, entries differing from the first solution %d (max %.3e), NaNs %d\n", n, reps, batch, fs[0], ndiff, dmax, nnan);
    // host reference: the reduced system (sig + E'E) x = r through T^-1 R (dense, n x n)
    if (with_ref) {
        std::vector<double> T((size_t)n * n, 0.0), R((size_t)n * n, 0.0);
        const double* SC = &vec[(size_t)V_SC * n];
        for (int m = 0; m < n; ++m) {
            const int m1 = (m + n - 1) % n, p1 = (m + 1) % n;
            const double s1 = SC[m1], s0 = SC[m];
            T[(size_t)m * n + m1] += 1.0; T[(size_t)m * n + m] += 2.0 * s1 * s1 + 2.0 * s1; T[(size_t)m * n + p1] += s1 * s0 * s0;
            R[(size_t)m * n + m1] += 3.0; R[(size_t)m * n + m] += -3.0 * (s1 + 1.0); R[(size_t)m * n + p1] += 3.0 * s1;
        }
        // D = T^-1 R by Gaussian elimination with partial pivoting on [T | R]
        std::vector<double> A(T), Bm(R);
        for (int k = 0; k < n; ++k) {
            int p = k;
            for (int i = k + 1; i < n; ++i) if (fabs(A[(size_t)i * n + k]) > fabs(A[(size_t)p * n + k])) p = i;
            if (p != k) for (int j = 0; j < n; ++j) { std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]); std::swap(Bm[(size_t)k * n + j], Bm[(size_t)p * n + j]); }
            const double inv = 1.0 / A[(size_t)k * n + k];
            for (int i = 0; i < n; ++i) {
                if (i == k) continue;
                const double f = A[(size_t)i * n + k] * inv;
                if (f == 0.0) continue;
                for (int j = k; j < n; ++j) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
                for (int j = 0; j < n; ++j) Bm[(size_t)i * n + j] -= f * Bm[(size_t)k * n + j];
            }
        }
        for (int k = 0; k < n; ++k) { const double inv = 1.0 / A[(size_t)k * n + k]; for (int j = 0; j < n; ++j) Bm[(size_t)k * n + j] *= inv; }
        // E = a D Nx + b D Ny
        std::vector<double> E((size_t)n * n);
        for (int i = 0; i < n; ++i) {
            const double cp = vec[(size_t)V_CP * n + i], a = -2.0 * cp * vec[(size_t)V_YP * n + i], b = 2.0 * cp * vec[(size_t)V_XP * n + i];
            for (int j = 0; j < n; ++j) E[(size_t)i * n + j] = pin[j] ? 0.0 : Bm[(size_t)i * n + j] * (a * vec[(size_t)V_NX * n + j] + b * vec[(size_t)V_NY * n + j]);
        }
        // residual of the GPU solution:  r - (sig x + E'(E x)),  relative to |sig x| + |E'||E x| + |r|
        std::vector<double> Ex(n, 0.0), res(n), scl(n);
        const double* x = out.data();
        for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < n; ++j) s += E[(size_t)i * n + j] * x[j]; Ex[i] = s; }
        double rmax = 0.0, smax = 0.0;
        for (int j = 0; j < n; ++j) {
            double s = 0.0, sa = 0.0;
            for (int i = 0; i < n; ++i) { s += E[(size_t)i * n + j] * Ex[i]; sa += fabs(E[(size_t)i * n + j] * Ex[i]); }
            const double sg = pin[j] ? 1.0 : vec[(size_t)V_SIG * n + j];
            rmax = fmax(rmax, fabs(rhs[j] - sg * x[j] - s));
            smax = fmax(smax, fabs(sg * x[j]) + sa + fabs(rhs[j]));
        }
        printf("   backward error of the first solution against the dense reduced system: %.3e\n", rmax / smax);
    }
    return (ndiff || nnan) ? 2 : 0;
}
