// mcq_kernels.hip -- see mcq_kernels.h for the overview and memory layouts.
#include "mcq_kernels.h"

#include <math.h>

// ONE translation unit (round 4): the solver's linear algebra is the saddle-point elimination of mcq_kkt.inc (minimum curvature) and the
// scalar cyclic-tridiagonal elimination of mcq_tri.inc (shortest path).  Compiled for TWO workgroups per CU: <= 256 VGPRs
// (hipcc --gpu-max-threads-per-block=512 + __launch_bounds__(256, 2)), 78 KB of LDS.


#define MCQ_NT 256
#define MCQ_NW (MCQ_NT / 64)
typedef double d2 __attribute__((vector_size(16)));
typedef __attribute__((address_space(1))) d2 gd2;

// ---------------------------------------------------------------------------------------------------------------------
// shared-memory carve-up of the solver kernel (doubles), saddle-point core: reduction scratch, the separators' system (persistent between
// a factorisation and its solves), the overlay (chunk buffers of the chains / scratch of the factorisation / the LDS copy of the
// curvature rows' Schur matrix), the curvature rows' lists.  78 KB: two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
#define SM_RED 0
#define SM_S (SM_RED + 64)
#define SPK 1920                                  /* mcq_kkt.inc: KP_* */
#define SM_OVL (SM_S + SPK)
#define OVL_SIZE 7464                             /* mcq_kkt.inc: KO_* (static_assert there) */
#define SM_KV (SM_OVL + OVL_SIZE)
#define SM_KI (SM_KV + 3 * MCQ_KMAX)
#define SM_TOTAL (SM_KI + (3 * MCQ_KMAX + 2 + 1) / 2 + 1)
static_assert(sizeof(double) * SM_TOTAL <= 80 * 1024, "two workgroups of the solver kernel per CU");

size_t mcq_solve_lds_bytes() { return sizeof(double) * SM_TOTAL; }

// The solver kernel's LDS: one statically sized array (address space 3 by type -> ds_read/ds_write in every device
// function, inlined or not).  78 KiB of the CU's 160 KiB: two workgroups per CU.
__shared__ double g_sm[SM_TOTAL];

// ---------------------------------------------------------------------------------------------------------------------
// small helpers
#ifndef MCQ_IPM_GM_MIN
#define MCQ_IPM_GM_MIN 0.995   /* interior point: fraction of the way to the boundary, max(GM_MIN, 1 - GM_C * mu / mu_0) */
#endif
#ifndef MCQ_IPM_GM_C
#define MCQ_IPM_GM_C 10.0
#endif
#ifndef MCQ_TAPIA_RATIO
#define MCQ_TAPIA_RATIO 0.7    /* Tapia evidence for an active row: s+/s < RATIO * z+/z and s+/s < SHRINK in the last interior-point step */
#endif
#ifndef MCQ_TAPIA_SHRINK
#define MCQ_TAPIA_SHRINK 0.7
#endif
#ifndef MCQ_WARM_ROUNDS
#define MCQ_WARM_ROUNDS 12
#endif
/* ^ rounds a warm-started exchange gets before the cold path (interior point) takes over: 8 / 10 / 12 / 16 measured on the
 *   3 x 1024 IQP problems -- the launch ends with its slowest problems, and a fallback costs the rounds spent plus the cold path */
#define MCQ_AS_WINDOW 8   /* block pivoting pins the furthest-out row per neighbourhood of this many rows either side */

// ---------------------------------------------------------------------------------------------------------------------
// i mod n for i in [-n, 2n): two compares instead of an integer division (indices one band width around the ring)
__device__ __forceinline__ int cyc1(int i, int n)
{
    i = i < 0 ? i + n : i;
    return i >= n ? i - n : i;
}

__device__ __forceinline__ int cyc(int i, int n)
{
    i %= n;
    return i < 0 ? i + n : i;
}

// signed cyclic difference a - b in (-n/2, n/2]
__device__ __forceinline__ int sdiff(int a, int b, int n)
{
    int d = cyc(a - b, n);
    return d > n / 2 ? d - n : d;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

// op: 0 sum, 1 min, 2 max.  All threads of the block must call; result returned to every thread.
__device__ double block_reduce_(double v, int op, double* red)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v = op == 0 ? wave_sum(v) : (op == 1 ? wave_min(v) : wave_max(v));
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double r = red[0];
    for (int k = 1; k < MCQ_NW; ++k) r = op == 0 ? r + red[k] : (op == 1 ? fmin(r, red[k]) : fmax(r, red[k]));
    return r;
}

// Two reductions (ops as above) for the price of one barrier pair; the wave-level butterflies interleave.
__device__ void block_reduce2_(double& a, int opa, double& b, int opb, double* red)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int m = 32; m >= 1; m >>= 1) {
        const double oa = __shfl_xor(a, m), ob = __shfl_xor(b, m);
        a = opa == 0 ? a + oa : (opa == 1 ? fmin(a, oa) : fmax(a, oa));
        b = opb == 0 ? b + ob : (opb == 1 ? fmin(b, ob) : fmax(b, ob));
    }
    __syncthreads();
    if (lane == 0) { red[wv] = a; red[MCQ_NW + wv] = b; }
    __syncthreads();
    double ra = red[0], rb = red[MCQ_NW];
    for (int k = 1; k < MCQ_NW; ++k) {
        ra = opa == 0 ? ra + red[k] : (opa == 1 ? fmin(ra, red[k]) : fmax(ra, red[k]));
        rb = opb == 0 ? rb + red[MCQ_NW + k] : (opb == 1 ? fmin(rb, red[MCQ_NW + k]) : fmax(rb, red[MCQ_NW + k]));
    }
    a = ra;
    b = rb;
}

// The inputs of one QP pass: the launch's own (McqBatch) for every kernel but mcq_iqp_rounds_kernel, which alternates between the two ring
// buffers of iqp_handler's rounds inside one launch.
struct McqSet {
    const int* n_list;
    const double* ref;
    const double* nv;
    const double* sc;
    const signed char* warm;
};
__device__ __forceinline__ McqSet mcq_set_of(const McqBatch& B)
{
    McqSet s;
    s.n_list = B.n_list; s.ref = B.ref; s.nv = B.nv; s.sc = B.sc; s.warm = B.warm;
    return s;
}

__device__ __forceinline__ McqWork mcq_work(const McqBatch& B, const McqSet& S, int pb, int& n, double& kb, double& wv)
{
    McqWork w;
    const size_t nm = (size_t)B.nmax;
    n = S.n_list ? S.n_list[pb] : B.n;
    kb = B.kappa_bound_list ? B.kappa_bound_list[pb] : B.kappa_bound;
    wv = B.w_veh_list ? B.w_veh_list[pb] : B.w_veh;
    w.ref = (const gdouble*)(S.ref + (size_t)pb * nm * 4);
    w.nv = S.nv ? (const gdouble*)(S.nv + (size_t)pb * nm * 2) : nullptr;
    w.sc = S.sc ? (const gdouble*)(S.sc + (size_t)pb * nm) : nullptr;
    w.L = (gdouble*)(B.L + (size_t)pb * nm * MCQ_LLD);
    w.vec = (gdouble*)(B.vec + (size_t)pb * nm * MCQ_NVEC);
    w.state = (gschar*)(B.state + (size_t)pb * nm);
    w.Z = (gdouble*)(B.Z + (size_t)pb * MCQ_ZLD(nm));
    w.alpha = B.alpha ? (gdouble*)(B.alpha + (size_t)pb * nm) : nullptr;
    w.curv_err = B.curv_err ? (gdouble*)(B.curv_err + pb) : nullptr;
    w.status = (gint*)(B.status + pb);
    w.info = B.info ? (ginfo*)(B.info + pb) : nullptr;
    return w;
}
__device__ __forceinline__ McqWork mcq_work(const McqBatch& B, int pb, int& n, double& kb, double& wv)
{
    return mcq_work(B, mcq_set_of(B), pb, n, kb, wv);
}

#define VEC(w, nmax, id) ((w).vec + (size_t)(id) * (size_t)(nmax))

// Broadcast of lane N of every 16-lane row to the whole row: one v_mov_b64_dpp row_newbcast (full-rate VALU, no SGPR round trip).
template <int N> __device__ __forceinline__ double bcast_row16(double v)
{
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xf, 0xf, true);
}
// c - (a of lane N of the 16-lane row) * b  as ONE instruction: the broadcast rides on the multiply-add as its DPP operand (fp64 VALU
// operations take row_newbcast).  hipcc 7.2 folds only 32-bit DPP moves into their users, hence the instruction by name; s_nop 1 = the two
// wait states a DPP read needs after a VALU write of the same register, which nobody checks inside an asm statement.  (The SIMT
// interpreter of tests/emu supplies this primitive itself, as it supplies the builtins.)
#ifndef MCQ_HAVE_FNMA_BCAST_ROW16
template <int N> __device__ __forceinline__ double fnma_bcast_row16(double a, double b, double c)
{
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b), "n"(N));
    return c;
}
#endif

#ifndef MCQ_IPM_TOL
#define MCQ_IPM_TOL 1e-10
#endif
// Everything a problem's workgroup shares that is not a vector: ONE instance in LDS (round 4).  Rounds 1-3 kept it as a private
// variable of the kernel and handed references to the non-inlined device functions: it then lives in scratch memory, every field access
// in a callee is a FLAT load (268 of them in the round-3 ISA), and with it went the kernel's arguments and the scalars of the solve.  As
// an LDS object -- the callees take `const LCtx&`, a reference that carries the address space -- a field is one ds_read.  Written by
// thread 0 (or by every thread with the same value) and published by the barriers that follow; the timers are thread 0's.
struct SolveCtx {
    McqDims d;
    McqWork w;
    int nm;
    int direct;                // 1: shortest-path objective -- H is a cyclic tridiagonal given entry by entry (V_SPD, V_SPU), V_F holds f
    int max_ipm_iter, max_as_iter, refine_steps;      // mcq_opts, as resolved by the host
    double zscale, fscale, wmean, nfree, kbound;      // scales of the tolerances (set by the kernel's prologue)
    mutable long long tk[8];   // phase timers (wall_clock64 ticks), thread 0's
    mutable int refine_rounds, second_attempt, f32_count;   // diagnostics for mcq_info
    mutable double last_step;  // length of the last interior-point step (the Tapia indicators are only trusted after a near-full one)
    mutable const gdouble* sp_sig;  // shortest path: the current "factorisation" = the diagonal shift and the working set (mcq_tri.inc, factor_sp)
    mutable const gschar* sp_mk;
    mutable const gdouble* kkt_w;   // saddle-point elimination: weights of the curvature rows (1 + y/t of the interior point) or nullptr
    mutable int kkt_f32;            // ... its records are stored as floats (mcq_kkt.inc, KRec): set by factor(), read by the solves that follow
    mutable int out_iters, out_nk;  // results of ipm / ipm_box / active_set besides their status (uniform: every thread writes the same)
    mutable double out_kkt;
};
typedef __attribute__((address_space(3))) SolveCtx LCtx;
__shared__ SolveCtx g_ctx;
#define G_CTX (*(const LCtx*)&g_ctx)
#define TICK() ((long long)wall_clock64())
#include "mcq_kkt.inc"
#include "mcq_tri.inc"

// =====================================================================================================================
// K1: assembly
// =====================================================================================================================
// Round 4: the assembly of a problem is a device function.  The solver kernel runs it as its prologue (one launch per QP pass; the
// spline quantities go from the assembly to the solver through vectors that are still in the L2 of the same workgroup's XCD), and
// mcq_assemble_kernel is the same function on its own for mcq_prep_device.  Returns the status it has written (uniform).
__device__ __noinline__ int assemble_problem(const LCtx& c, double wveh, gdouble* nv_out, gdouble* sc_out)
{
    double* red = g_sm + SM_RED;
    const int tid = threadIdx.x;
    McqWork w;                    // (member by member: a struct in LDS has no implicit copy into a private one)
    w.ref = c.w.ref; w.nv = c.w.nv; w.sc = c.w.sc; w.L = c.w.L; w.vec = c.w.vec; w.state = c.w.state; w.Z = c.w.Z;
    w.alpha = c.w.alpha; w.curv_err = c.w.curv_err; w.status = c.w.status; w.info = c.w.info;
    const int nm = c.nm, n = c.d.n;
    gdouble* LO = VEC(w, nm, V_LO);
    gdouble* HI = VEC(w, nm, V_HI);
    gdouble* S = VEC(w, nm, V_SC);    // spline scalings
    gdouble* XP = VEC(w, nm, V_XP);
    gdouble* YP = VEC(w, nm, V_YP);
    gdouble* CP = VEC(w, nm, V_CP);
    gdouble* KRF = VEC(w, nm, V_KREF);
    gdouble* XPP = VEC(w, nm, V_XPP);
    gdouble* YPP = VEC(w, nm, V_YPP);

    // ---- phase 0: validate, box bounds  [-(w_l - w_veh/2), w_r - w_veh/2]   -----------------
    double flag_bad = 0.0, flag_inf = 0.0;
    if (n < 3) flag_bad = 1.0;
    gdouble* NX = VEC(w, nm, V_NX);
    gdouble* NY = VEC(w, nm, V_NY);
    const bool derive = w.nv == nullptr;      // normals and scalings from the distance-scaled spline through the line itself
    for (int i = tid; i < n; i += MCQ_NT) {
        const double x = w.ref[4 * i], y = w.ref[4 * i + 1], wr = w.ref[4 * i + 2], wl = w.ref[4 * i + 3];
        const double nx = derive ? 0.0 : w.nv[2 * i], ny = derive ? 1.0 : w.nv[2 * i + 1];
        double s = w.sc ? w.sc[i] : 1.0;
        if (derive) {
            // s_i = l_i / l_{i+1},  l_i = |p_{i+1} - p_i|  (tph.calc_splines, use_dist_scaling=True, closed)
            const int i1 = i + 1 >= n ? i + 1 - n : i + 1, i2 = i1 + 1 >= n ? i1 + 1 - n : i1 + 1;
            const double l0 = hypot(w.ref[4 * i1] - x, w.ref[4 * i1 + 1] - y);
            const double l1 = hypot(w.ref[4 * i2] - w.ref[4 * i1], w.ref[4 * i2 + 1] - w.ref[4 * i1 + 1]);
            s = l0 / l1;
        } else {
            NX[i] = nx;
            NY[i] = ny;
        }
        if (!(isfinite(x) && isfinite(y) && isfinite(wr) && isfinite(wl) && isfinite(nx) && isfinite(ny) && isfinite(s)
              && s > 0.0))
            flag_bad = 1.0;
        const double lo = -(wl - 0.5 * wveh), hi = wr - 0.5 * wveh;
        if (hi < lo) flag_inf = 1.0;
        LO[i] = lo;
        HI[i] = hi;
        S[i] = s;
    }
    flag_bad = block_reduce_(flag_bad, 2, red);
    flag_inf = block_reduce_(flag_inf, 2, red);
    const int st = flag_bad > 0.0 ? MCQ_BAD_INPUT : (flag_inf > 0.0 ? MCQ_INFEASIBLE : MCQ_OK);
    if (tid == 0) {
        *w.status = st;
        if (w.curv_err) *w.curv_err = 0.0;
        if (w.info) {
            mcq_info z;
            z.ipm_iters = z.as_iters = z.n_active_box = z.n_active_kappa = 0;
            z.kappa_max = 0.0;
            z.kkt_res = 0.0;
            z.refine_rounds = z.second_attempt = z.f32_factorisations = z.gi_iters = 0;
            for (int q = 0; q < 8; ++q) z.ticks[q] = 0;
            *(mcq_info*)w.info = z;
        }
    }
    if (st != MCQ_OK) {
        if (w.alpha) for (int i = tid; i < n; i += MCQ_NT) w.alpha[i] = 0.0;
        return st;
    }
    __syncthreads();

    // ---- phases 1-2: second derivatives of the closed spline through the reference line.  The c-coefficients solve the cyclic tridiagonal
    //      system   1 c_(m-1) + (2 s_(m-1)^2 + 2 s_(m-1)) c_m + (s_(m-1) s_m^2) c_(m+1) = 3 (s_(m-1) D_m - D_(m-1)),   D_m = p_(m+1) - p_m
    //      (SURVEY.md App. A.1); x'' = 2 c.  Solved by the periodic-pivot sweeps of mcq_tri.inc, whose factors (V_IDL, V_TUC) the solver
    //      kernel goes on using.  (Rounds 1-3 built rows of T^-1 here -- with image folding for short rings -- and, from them, the bands
    //      of E, E' and D; nothing reads a band any more.)
    tri_prepare(c);
    gdouble* RX = VEC(w, nm, V_SK);
    gdouble* RY = VEC(w, nm, V_EDA);
    for (int m = tid; m < n; m += MCQ_NT) {
        const int mp = cyc1(m + 1, n), mm = cyc1(m - 1, n);
        const double sm1 = S[mm];
        RX[m] = 3.0 * (sm1 * (w.ref[4 * mp] - w.ref[4 * m]) - (w.ref[4 * m] - w.ref[4 * mm]));
        RY[m] = 3.0 * (sm1 * (w.ref[4 * mp + 1] - w.ref[4 * m + 1]) - (w.ref[4 * m + 1] - w.ref[4 * mm + 1]));
    }
    tri_solve_T(c, RX, XPP);
    tri_solve_T(c, RY, YPP);
    for (int i = tid; i < n; i += MCQ_NT) { XPP[i] *= 2.0; YPP[i] *= 2.0; }     // x''(0), y''(0) of spline i
    __syncthreads();

    // ---- phase 3a: x', y', curvature pre-factor, reference curvature ------------------------------------------------
    for (int i = tid; i < n; i += MCQ_NT) {
        const int ip = cyc(i + 1, n);
        const double s2 = S[i] * S[i];
        const double xp = (w.ref[4 * ip] - w.ref[4 * i]) - (XPP[i] + 0.5 * s2 * XPP[ip]) / 3.0;
        const double yp = (w.ref[4 * ip + 1] - w.ref[4 * i + 1]) - (YPP[i] + 0.5 * s2 * YPP[ip]) / 3.0;
        const double den = pow(xp * xp + yp * yp, 1.5);
        const double cp = den != 0.0 ? 1.0 / den : 0.0;
        XP[i] = xp;
        YP[i] = yp;
        CP[i] = cp;
        KRF[i] = cp * (xp * YPP[i] - yp * XPP[i]);
        if (derive) {
            // unit normal to the right of the spline's tangent (b-coefficients):  (y', -x') / |.|   (tph.calc_splines)
            const double nrm = sqrt(xp * xp + yp * yp);
            NX[i] = yp / nrm;
            NY[i] = -xp / nrm;
        }
        if (nv_out) {
            nv_out[2 * i] = NX[i];
            nv_out[2 * i + 1] = NY[i];
        }
        if (sc_out) sc_out[i] = S[i];
    }
    __syncthreads();
    return MCQ_OK;
}

// what the assembly needs of the context: the problem's pointers and size (thread 0 writes, the barrier publishes)
__device__ __forceinline__ void ctx_set_problem(const McqBatch& B, const McqSet& S, int pb, int& n, double& kb, double& wveh)
{
    const McqWork w = mcq_work(B, S, pb, n, kb, wveh);
    __syncthreads();                                  // (a previous use of the context by this workgroup -- none today -- is over)
    if (threadIdx.x == 0) {
        g_ctx.w = w;
        g_ctx.nm = B.nmax;
        g_ctx.d = mcq_dims(n);
    }
    __syncthreads();
}

__device__ __forceinline__ void ctx_set_problem(const McqBatch& B, int pb, int& n, double& kb, double& wveh)
{
    ctx_set_problem(B, mcq_set_of(B), pb, n, kb, wveh);
}

// ... and what a solve needs on top: the options, the timers and diagnostics at zero
__device__ __forceinline__ void ctx_set_solve(const McqBatch& B, double kbound)
{
    if (threadIdx.x == 0) {
        for (int q = 0; q < 8; ++q) g_ctx.tk[q] = 0;
        g_ctx.last_step = 0.0;
        g_ctx.refine_rounds = g_ctx.second_attempt = g_ctx.f32_count = 0;
        g_ctx.direct = B.objective == MCQ_OBJ_SHORTEST_PATH;
        g_ctx.max_ipm_iter = B.max_ipm_iter;
        g_ctx.max_as_iter = B.max_as_iter;
        g_ctx.refine_steps = B.refine_steps;
        g_ctx.kbound = kbound;
        g_ctx.kkt_w = nullptr;
        g_ctx.kkt_f32 = 0;
        g_ctx.sp_sig = nullptr;
        g_ctx.sp_mk = nullptr;
        g_ctx.out_iters = g_ctx.out_nk = 0;
        g_ctx.out_kkt = 0.0;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(MCQ_NT) mcq_assemble_kernel(McqBatch B)
{
    int n;
    double kb, wveh;
    ctx_set_problem(B, blockIdx.x, n, kb, wveh);
    const size_t nm = (size_t)B.nmax;
    (void)assemble_problem(G_CTX, wveh, B.nv_out ? (gdouble*)(B.nv_out + (size_t)blockIdx.x * nm * 2) : nullptr,
                           B.sc_out ? (gdouble*)(B.sc_out + (size_t)blockIdx.x * nm) : nullptr);
}

// =====================================================================================================================
// K3: solver
// =====================================================================================================================
// =====================================================================================================================
// K1': assembly of the shortest-path QP (tph.opt_shortest_path, call site
//      [REF main_globaltraj.py:286-290]):   minimise  sum_i |p_{i+1} + a_{i+1} n_{i+1} - p_i - a_i n_i|^2   over the ring,
//      i.e.  1/2 a'Ha + f'a  with  H_ii = 4 |n_i|^2,  H_{i,i+1} = -2 n_i . n_{i+1},  f_i = 2 n_i . (2 p_i - p_{i-1} - p_{i+1}),
//      on the box  -max(w_l - w_veh/2, 0.001) <= a_i <= max(w_r - w_veh/2, 0.001).
//      H is kept as two vectors (diagonal, H[i, i + 1 mod n]): the solver kernel's scalar cyclic-tridiagonal route (mcq_tri.inc).
// =====================================================================================================================
__global__ void __launch_bounds__(MCQ_NT) mcq_assemble_sp_kernel(McqBatch B)
{
    __shared__ double red[64];
    const int tid = threadIdx.x;
    int n;
    double kb, wveh;
    const McqWork w = mcq_work(B, blockIdx.x, n, kb, wveh);
    const int nm = B.nmax;
    gdouble* LO = VEC(w, nm, V_LO);
    gdouble* HI = VEC(w, nm, V_HI);
    gdouble* NX = VEC(w, nm, V_NX);
    gdouble* NY = VEC(w, nm, V_NY);
    gdouble* F = VEC(w, nm, V_F);
    gdouble* KRF = VEC(w, nm, V_KREF);

    double flag_bad = n < 3 ? 1.0 : 0.0;
    for (int i = tid; i < n; i += MCQ_NT) {
        const double x = w.ref[4 * i], y = w.ref[4 * i + 1], wr = w.ref[4 * i + 2], wl = w.ref[4 * i + 3];
        const double nx = w.nv[2 * i], ny = w.nv[2 * i + 1];
        if (!(isfinite(x) && isfinite(y) && isfinite(wr) && isfinite(wl) && isfinite(nx) && isfinite(ny))) flag_bad = 1.0;
        NX[i] = nx;
        NY[i] = ny;
        LO[i] = -fmax(wl - 0.5 * wveh, 0.001);
        HI[i] = fmax(wr - 0.5 * wveh, 0.001);
        KRF[i] = 0.0;
    }
    flag_bad = block_reduce_(flag_bad, 2, red);
    const int st = flag_bad > 0.0 ? MCQ_BAD_INPUT : MCQ_OK;
    if (tid == 0) {
        *w.status = st;
        if (w.curv_err) *w.curv_err = 0.0;
        if (w.info) {
            mcq_info z;
            z.ipm_iters = z.as_iters = z.n_active_box = z.n_active_kappa = 0;
            z.kappa_max = 0.0;
            z.kkt_res = 0.0;
            z.refine_rounds = z.second_attempt = z.f32_factorisations = z.gi_iters = 0;
            for (int q = 0; q < 8; ++q) z.ticks[q] = 0;
            *(mcq_info*)w.info = z;
        }
    }
    if (st != MCQ_OK) {
        if (w.alpha) for (int i = tid; i < n; i += MCQ_NT) w.alpha[i] = 0.0;
        return;
    }

    // entries of the cyclic tridiagonal (n >= 3: the two neighbours of a point are distinct)
    gdouble* HD = VEC(w, nm, V_XP);       // V_SPD / V_SPU of mcq_tri.inc
    gdouble* HU = VEC(w, nm, V_YP);
    for (int i = tid; i < n; i += MCQ_NT) {
        const int ip = i + 1 == n ? 0 : i + 1, im = i == 0 ? n - 1 : i - 1;
        HD[i] = 4.0 * (w.nv[2 * i] * w.nv[2 * i] + w.nv[2 * i + 1] * w.nv[2 * i + 1]);
        HU[i] = -2.0 * (w.nv[2 * i] * w.nv[2 * ip] + w.nv[2 * i + 1] * w.nv[2 * ip + 1]);
        const double px = w.ref[4 * i], py = w.ref[4 * i + 1];
        F[i] = 2.0 * (w.nv[2 * i] * ((px - w.ref[4 * im]) - (w.ref[4 * ip] - px))
                      + w.nv[2 * i + 1] * ((py - w.ref[4 * im + 1]) - (w.ref[4 * ip + 1] - py)));
    }
}


// fv: the right-hand side of the solve that follows rides through the elimination (solve(c, fv, true) then finishes it);
// f32: the records of this factorisation may be stored as floats (interior-point iterations: inexact Newton directions, see KRec)
__device__ __noinline__ int factor(const LCtx& c, const gdouble* sig, const gschar* mk, gdouble* fv, bool f32)
{
    if (c.direct) return factor_sp(c, sig, mk);
    c.kkt_f32 = f32 ? 1 : 0;
    if (threadIdx.x == 0) c.f32_count += f32 ? 1 : 0;
    return f32 ? factor_kkt<true>(c, sig, mk, c.kkt_w, fv) : factor_kkt<false>(c, sig, mk, c.kkt_w, fv);
}

__device__ __noinline__ void solve(const LCtx& c, gdouble* v, bool fwd_done)
{
    if (c.direct) { (void)sp_solve(c, v, v); return; }
    if (c.kkt_f32) solve_kkt<true>(c, v, fwd_done);
    else solve_kkt<false>(c, v, fwd_done);
}

// dst = E' src, through the spline system (mcq_tri.inc)
__device__ __forceinline__ void apply_Et(const LCtx& c, const gdouble* src, gdouble* dst) { tri_apply_Et(c, src, dst); }

// g = E'(E x + F_SCALE k_ref + extra)      (tmp: scratch vector; extra may be nullptr)
__device__ __noinline__ void gradient(const LCtx& c, const gdouble* x, const gdouble* extra, gdouble* tmp, gdouble* g)
{
    const int n = c.d.n;
    const long long t0 = TICK();
    __syncthreads();
    if (c.direct) {          // g = H x + f, H a cyclic tridiagonal given entry by entry
        const gdouble* HD = VEC(c.w, c.nm, V_SPD);
        const gdouble* HU = VEC(c.w, c.nm, V_SPU);
        const gdouble* F = VEC(c.w, c.nm, V_F);
        for (int i = threadIdx.x; i < n; i += MCQ_NT) {
            const int im = i > 0 ? i - 1 : n - 1, ip = i + 1 < n ? i + 1 : 0;
            g[i] = F[i] + HU[im] * x[im] + HD[i] * x[i] + HU[i] * x[ip];
        }
        __syncthreads();
        if (threadIdx.x == 0) c.tk[2] += TICK() - t0;
        return;
    }
    // E and E' through the spline system itself (mcq_tri.inc): no band is read
    tri_apply_E(c, x, VEC(c.w, c.nm, V_KREF), MCQ_F_SCALE, tmp);
    if (extra) {
        for (int i = threadIdx.x; i < n; i += MCQ_NT) tmp[i] += extra[i];
        __syncthreads();
    }
    tri_apply_Et(c, tmp, g);
    if (threadIdx.x == 0) c.tk[2] += TICK() - t0;
}

__device__ __forceinline__ int timed_factor(const LCtx& c, const gdouble* sig, const gschar* mk, gdouble* fv = nullptr, bool f32 = false)
{
    const long long t0 = TICK();
    const int r = factor(c, sig, mk, fv, f32);
    if (threadIdx.x == 0) c.tk[0] += TICK() - t0;
    return r;
}
__device__ __forceinline__ void timed_solve(const LCtx& c, gdouble* v, bool fwd_done = false)
{
    const long long t0 = TICK();
    solve(c, v, fwd_done);
    if (threadIdx.x == 0) c.tk[1] += TICK() - t0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Mehrotra predictor-corrector interior point on
//     min 1/2 x'Hx + f'x   s.t.  lo <= x <= hi   [ and, with_kappa:  -kb <= E x + k_ref <= kb ]
// Box pairs (sl, zl), (su, zu); curvature pairs (tl, yl), (tu, yu) with tl = kb + r, tu = kb - r, r = E x + k_ref
// carried as infeasible-start slacks (residuals rho).  The reduced system is
//     (H + diag(zl/sl + zu/su) + E' diag(yl/tl + yu/tu) E) dx = rhs
// i.e. the same bordered band as H: one banded Cholesky per iteration, two solves.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __noinline__ int ipm(const LCtx& c, bool with_kappa)
{
    const int tid = threadIdx.x, n = c.d.n, nm = c.nm;
    double* red = g_sm + SM_RED;
    const gdouble* LO = VEC(c.w, nm, V_LO);
    const gdouble* HI = VEC(c.w, nm, V_HI);
    const gdouble* KR = VEC(c.w, nm, V_KREF);
    gdouble* X = VEC(c.w, nm, V_X);
    gdouble* G = VEC(c.w, nm, V_G);
    gdouble* ZL = VEC(c.w, nm, V_ZL);
    gdouble* ZU = VEC(c.w, nm, V_ZU);
    gdouble* SIG = VEC(c.w, nm, V_SIG);
    gdouble* RHS = VEC(c.w, nm, V_RHS);
    gdouble* DXA = VEC(c.w, nm, V_DXA);
    gdouble* T0 = VEC(c.w, nm, V_T0);
    gdouble* T1 = VEC(c.w, nm, V_T1);
    gdouble* T2 = VEC(c.w, nm, V_T2);
    gdouble* TL = VEC(c.w, nm, V_TL);
    gdouble* TU = VEC(c.w, nm, V_TU);
    gdouble* YL = VEC(c.w, nm, V_YL);
    gdouble* YU = VEC(c.w, nm, V_YU);
    gdouble* SK = VEC(c.w, nm, V_SK);
    gdouble* EDA = VEC(c.w, nm, V_EDA);
    gdouble* Q = VEC(c.w, nm, V_Q);
    gschar* ST = c.w.state;
    const double kb = c.kbound, zscale = c.zscale;
    const double IPM_TOL = 1e-10;
    c.out_iters = 0;

    for (int i = tid; i < n; i += MCQ_NT) {
        const bool fixed = !(HI[i] - LO[i] > 1e-12);
        ST[i] = fixed ? 2 : 0;
        X[i] = 0.5 * (LO[i] + HI[i]);
        ZL[i] = fixed ? 0.0 : zscale;
        ZU[i] = ZL[i];
    }
    __syncthreads();
    if (with_kappa) {
        tri_apply_E(c, X, KR, 1.0, T0);     // r = E x + k_ref
        const double mu0 = 0.5 * zscale * c.wmean;
        for (int i = tid; i < n; i += MCQ_NT) {
            const double r = T0[i];
            const double tl = fmax(kb + r, 0.1 * kb), tu = fmax(kb - r, 0.1 * kb);
            TL[i] = tl; TU[i] = tu;
            YL[i] = mu0 / tl; YU[i] = mu0 / tu;
        }
        __syncthreads();
    }
    const double npairs = 2.0 * c.nfree + (with_kappa ? 2.0 * n : 0.0);
    const gschar* const no_mask = nullptr;
    const bool any_fixed = c.nfree < (double)n;
    (void)no_mask;
    if (!(npairs > 0.0)) return MCQ_OK;

    // Box-only phase: the gradient g = H x + f is carried along instead of recomputed.  The reduced system gives
    //   H dx = rhs - diag(sig) dx   on the free rows (pinned rows have dx = 0 and their g is never read),
    // so  g(x + a dx) = g + a (rhs - sig dx)  costs a vector pass where two band products (2 MB of E / E') were; the error
    // it carries is the residual of the banded solve.  Convergence is only declared on an exactly recomputed gradient.
    // On entry G holds the exact gradient at the box centre (computed by the caller for the scaling).
    bool g_exact = true;
    double mu_first = 0.0, rdm_best = 1e300;
    for (int it = 1; it <= c.max_ipm_iter; ++it) {
        // ---- residuals: g = H x + f;  with kappa also r, rho and the dual residual needs E'(yu - yl) ---------------------
        if (with_kappa) {
            for (int i = tid; i < n; i += MCQ_NT) Q[i] = YU[i] - YL[i];
            gradient(c, X, Q, T0, T1);                                   // T1 = g + E'(yu - yl)   (dual residual part)
            gradient(c, X, nullptr, T0, G);                              // G  = g
            tri_apply_E(c, X, KR, 1.0, T0);  // T0 = r
        }
        double mu, rdm, rhom;
        for (;;) {
            mu = 0.0; rdm = 0.0; rhom = 0.0;
            for (int i = tid; i < n; i += MCQ_NT) {
                if (with_kappa) {
                    mu += TL[i] * YL[i] + TU[i] * YU[i];
                    rhom = fmax(rhom, fmax(fabs(TL[i] - (kb + T0[i])), fabs(TU[i] - (kb - T0[i]))));
                    SK[i] = YL[i] / TL[i] + YU[i] / TU[i];
                }
                if (ST[i] != 0) continue;
                const double sl = X[i] - LO[i], su = HI[i] - X[i];
                mu += sl * ZL[i] + su * ZU[i];
                const double gg = with_kappa ? T1[i] : G[i];
                rdm = fmax(rdm, fabs(gg - ZL[i] + ZU[i]));
                SIG[i] = ZL[i] / sl + ZU[i] / su;
            }
            mu = block_reduce_(mu, 0, red) / npairs;
            rdm = block_reduce_(rdm, 2, red);
            rhom = block_reduce_(rhom, 2, red);
            const bool conv = mu < IPM_TOL * zscale * c.wmean && rdm < IPM_TOL * zscale && rhom <= 1e-9 * kb;
            // Curvature rows, late in the path: the weights 1 + y / t of rows close to their bound reach 1e10 and more, and the dual residual
            // stops following the complementarity down -- rounding in the weighted system, whichever way it is solved (the band of
            // E' diag(1 + y/t) E met a non-positive pivot at this point; the saddle-point form carries the weights in its (cx, cy) blocks and
            // loses the digits there).  Once the complementarity is down by 1e-6 and the dual residual has turned around, the pairs identify
            // the working set as well as they ever will: the exact active-set phase that follows does not use weights.
            if (with_kappa && it > 1 && mu < 1e-6 * mu_first && rdm > 10.0 * rdm_best) return MCQ_OK;
            rdm_best = fmin(rdm_best, rdm);
#ifdef IPM_TRACE
            if (tid == 0) printf("ipm%d it %d mu %.3e rdm %.3e rhom %.3e step %.3e\n", (int)with_kappa, it, mu / (zscale * c.wmean), rdm / zscale, rhom / kb, c.last_step);
#endif
            if (conv && (with_kappa || g_exact)) return MCQ_OK;
            if (!conv) break;
            gradient(c, X, nullptr, T0, G);          // looks converged on the carried gradient: confirm on the exact one
            g_exact = true;
        }
        c.out_iters = it;
        if (it == 1) mu_first = mu;

        // ---- factorisation of the reduced system ---------------------------------------------------------------------
        int fs;
        if (with_kappa) {
            // H slab <- E' (I + diag(SK)) E = H + E' diag(SK) E   (H = E'E is rebuilt by the caller after this phase)
            for (int i = tid; i < n; i += MCQ_NT) EDA[i] = 1.0 + SK[i];
            __syncthreads();
            c.kkt_w = EDA;                           // the weights enter the (cx, cy) block of every waypoint: nothing to form
            fs = timed_factor(c, SIG, any_fixed ? ST : nullptr);
            c.kkt_w = nullptr;
            // With many curvature rows close to their bound the weights y / t reach 1e10 and more near the end; the band of
            // E' diag(1 + SK) E then carries rounding errors of that size against eigenvalues of order one, and the Cholesky can
            // meet a non-positive pivot although the matrix is positive definite (round 3: a 360-point stadium with 134 active
            // rows).  Late in the path -- the complementarity is down by 1e-6 from where it started -- that is no reason to give
            // up: the pairs of the last completed iteration identify the working set, and the exact active-set phase that follows
            // does not form this matrix.
            if (fs == MCQ_NOT_PD && mu < 1e-6 * mu_first) return MCQ_OK;
        } else {
            fs = timed_factor(c, SIG, any_fixed ? ST : nullptr);
        }
        if (fs != 0) return fs;

        // ---- predictor -------------------------------------------------------------------------------------------------
        if (with_kappa) {
            // rhs = -g + E' q,  q = yl rho_l / tl - yu rho_u / tu
            for (int i = tid; i < n; i += MCQ_NT) {
                const double rl = TL[i] - (kb + T0[i]), ru = TU[i] - (kb - T0[i]);
                Q[i] = YL[i] * rl / TL[i] - YU[i] * ru / TU[i];
            }
            __syncthreads();
            apply_Et(c, Q, T2);
            __syncthreads();
            for (int i = tid; i < n; i += MCQ_NT) RHS[i] = ST[i] == 0 ? -G[i] + T2[i] : 0.0;
        } else {
            for (int i = tid; i < n; i += MCQ_NT) RHS[i] = ST[i] == 0 ? -G[i] : 0.0;
        }
        timed_solve(c, RHS);
        if (with_kappa) {
            tri_apply_E(c, RHS, nullptr, 0.0, EDA);   // E dx_aff
        }
        double ap = 1.0, ad = 1.0;
        for (int i = tid; i < n; i += MCQ_NT) {
            if (with_kappa) {
                const double rl = TL[i] - (kb + T0[i]), ru = TU[i] - (kb - T0[i]);
                const double dtl = EDA[i] - rl, dtu = -EDA[i] - ru;
                const double dyl = -YL[i] - YL[i] * dtl / TL[i], dyu = -YU[i] - YU[i] * dtu / TU[i];
                if (dtl < 0.0) ap = fmin(ap, -TL[i] / dtl);
                if (dtu < 0.0) ap = fmin(ap, -TU[i] / dtu);
                if (dyl < 0.0) ad = fmin(ad, -YL[i] / dyl);
                if (dyu < 0.0) ad = fmin(ad, -YU[i] / dyu);
            }
            if (ST[i] != 0) { DXA[i] = 0.0; continue; }
            const double dx = RHS[i];
            DXA[i] = dx;
            const double sl = X[i] - LO[i], su = HI[i] - X[i];
            const double dzl = -ZL[i] - ZL[i] * dx / sl, dzu = -ZU[i] + ZU[i] * dx / su;
            if (dx < 0.0) ap = fmin(ap, -sl / dx);
            if (dx > 0.0) ap = fmin(ap, su / dx);
            if (dzl < 0.0) ad = fmin(ad, -ZL[i] / dzl);
            if (dzu < 0.0) ad = fmin(ad, -ZU[i] / dzu);
        }
        ap = block_reduce_(ap, 1, red);
        ad = block_reduce_(ad, 1, red);
        double mua = 0.0;
        for (int i = tid; i < n; i += MCQ_NT) {
            if (with_kappa) {
                const double rl = TL[i] - (kb + T0[i]), ru = TU[i] - (kb - T0[i]);
                const double dtl = EDA[i] - rl, dtu = -EDA[i] - ru;
                const double dyl = -YL[i] - YL[i] * dtl / TL[i], dyu = -YU[i] - YU[i] * dtu / TU[i];
                mua += (TL[i] + ap * dtl) * (YL[i] + ad * dyl) + (TU[i] + ap * dtu) * (YU[i] + ad * dyu);
            }
            if (ST[i] != 0) continue;
            const double dx = DXA[i];
            const double sl = X[i] - LO[i], su = HI[i] - X[i];
            const double dzl = -ZL[i] - ZL[i] * dx / sl, dzu = -ZU[i] + ZU[i] * dx / su;
            mua += (sl + ap * dx) * (ZL[i] + ad * dzl) + (su - ap * dx) * (ZU[i] + ad * dzu);
        }
        mua = block_reduce_(mua, 0, red) / npairs;
        const double ratio = mua / mu;
        const double smu = ratio * ratio * ratio * mu;

        // ---- corrector -------------------------------------------------------------------------------------------------
        if (with_kappa) {
            for (int i = tid; i < n; i += MCQ_NT) {
                const double rl = TL[i] - (kb + T0[i]), ru = TU[i] - (kb - T0[i]);
                const double dtl = EDA[i] - rl, dtu = -EDA[i] - ru;
                const double dyl = -YL[i] - YL[i] * dtl / TL[i], dyu = -YU[i] - YU[i] * dtu / TU[i];
                Q[i] = (smu - dtl * dyl + YL[i] * rl) / TL[i] - (smu - dtu * dyu + YU[i] * ru) / TU[i];
            }
            __syncthreads();
            apply_Et(c, Q, T2);
            __syncthreads();
        }
        for (int i = tid; i < n; i += MCQ_NT) {
            if (ST[i] != 0) { RHS[i] = 0.0; continue; }
            const double dx = DXA[i];
            const double sl = X[i] - LO[i], su = HI[i] - X[i];
            const double dzl = -ZL[i] - ZL[i] * dx / sl, dzu = -ZU[i] + ZU[i] * dx / su;
            RHS[i] = -G[i] + (smu - dx * dzl) / sl - (smu + dx * dzu) / su + (with_kappa ? T2[i] : 0.0);
        }
        timed_solve(c, RHS);
        if (with_kappa) {
            tri_apply_E(c, RHS, nullptr, 0.0, Q);     // Q = E dx
        }
        double amax = 1.0 / 0.995;
        for (int i = tid; i < n; i += MCQ_NT) {
            if (with_kappa) {
                const double rl = TL[i] - (kb + T0[i]), ru = TU[i] - (kb - T0[i]);
                const double dtla = EDA[i] - rl, dtua = -EDA[i] - ru;
                const double dyla = -YL[i] - YL[i] * dtla / TL[i], dyua = -YU[i] - YU[i] * dtua / TU[i];
                const double dtl = Q[i] - rl, dtu = -Q[i] - ru;
                const double dyl = (-TL[i] * YL[i] + smu - dtla * dyla - YL[i] * dtl) / TL[i];
                const double dyu = (-TU[i] * YU[i] + smu - dtua * dyua - YU[i] * dtu) / TU[i];
                // stash the curvature-pair directions: EDA <- dyl, SK <- dyu (both are recomputed next iteration)
                EDA[i] = dyl;
                SK[i] = dyu;
                if (dtl < 0.0) amax = fmin(amax, -TL[i] / dtl);
                if (dtu < 0.0) amax = fmin(amax, -TU[i] / dtu);
                if (dyl < 0.0) amax = fmin(amax, -YL[i] / dyl);
                if (dyu < 0.0) amax = fmin(amax, -YU[i] / dyu);
            }
            if (ST[i] != 0) continue;
            const double dx = RHS[i], da = DXA[i];
            const double sl = X[i] - LO[i], su = HI[i] - X[i];
            const double dzla = -ZL[i] - ZL[i] * da / sl, dzua = -ZU[i] + ZU[i] * da / su;
            const double dzl = (-sl * ZL[i] + smu - da * dzla - ZL[i] * dx) / sl;
            const double dzu = (-su * ZU[i] + smu + da * dzua + ZU[i] * dx) / su;
            T1[i] = dzl;
            T2[i] = dzu;
            if (dx < 0.0) amax = fmin(amax, -sl / dx);
            if (dx > 0.0) amax = fmin(amax, su / dx);
            if (dzl < 0.0) amax = fmin(amax, -ZL[i] / dzl);
            if (dzu < 0.0) amax = fmin(amax, -ZU[i] / dzu);
        }
        amax = block_reduce_(amax, 1, red);
        const double a = fmin(1.0, 0.995 * amax);
        c.last_step = a;
        for (int i = tid; i < n; i += MCQ_NT) {
            if (with_kappa) {
                const double rl = TL[i] - (kb + T0[i]), ru = TU[i] - (kb - T0[i]);
                TL[i] += a * (Q[i] - rl);
                TU[i] += a * (-Q[i] - ru);
                YL[i] += a * EDA[i];
                YU[i] += a * SK[i];
            }
            if (ST[i] != 0) continue;
            if (!with_kappa) {
                // g += a H dx,  H dx = (corrector right-hand side) - sig dx
                const double dxa = DXA[i], dx = RHS[i];
                const double sl = X[i] - LO[i], su = HI[i] - X[i];
                const double dzla = -ZL[i] - ZL[i] * dxa / sl, dzua = -ZU[i] + ZU[i] * dxa / su;
                const double rc = -G[i] + (smu - dxa * dzla) / sl - (smu + dxa * dzua) / su;
                G[i] += a * (rc - SIG[i] * dx);
            }
            {
                const double sl = X[i] - LO[i], su = HI[i] - X[i];
                const double rsl = (sl + a * RHS[i]) * ZL[i], rzl = (ZL[i] + a * T1[i]) * sl;
                const double rsu = (su - a * RHS[i]) * ZU[i], rzu = (ZU[i] + a * T2[i]) * su;
                const bool al = rsl < MCQ_TAPIA_RATIO * rzl && sl + a * RHS[i] < MCQ_TAPIA_SHRINK * sl, au = rsu < MCQ_TAPIA_RATIO * rzu && su - a * RHS[i] < MCQ_TAPIA_SHRINK * su;
                VEC(c.w, nm, V_T3)[i] = al ? -1.0 : (au ? 1.0 : 0.0);                     // Tapia indicators, see active_set()
            }
            X[i] += a * RHS[i];
            ZL[i] += a * T1[i];
            ZU[i] += a * T2[i];
        }
        g_exact = false;
        __syncthreads();
    }
    return with_kappa ? MCQ_KAPPA_INFEASIBLE : MCQ_ITER_CAP;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same Mehrotra iteration for the box-only phase of rings with n <= 8 x 256, with the vector work restructured for a
// workgroup that runs one wave per SIMD: every thread owns its (up to) eight entries i = tid + 256 u, ALL loads of a pass
// are issued before the first use (one L2 round trip per pass instead of one per entry), and what a pass loaded stays in
// registers across the block reductions that follow it -- three load phases and six reductions per iteration where the
// generic routine above does seven dependent passes.  Arithmetic and summation order are those of ipm().
// ---------------------------------------------------------------------------------------------------------------------
#define IPB_E 8
#define IPB_H 4     /* entries a pass keeps in registers at a time: two halves per pass (round 3: at 256 VGPRs -- two workgroups per CU -- eight entries of a pass's arrays spilled) */
// `resume`: continue from the pairs (X, ZL, ZU) already in memory to the tighter tolerance `tol` -- the second attempt on
// degenerate / very ill-conditioned instances (see the driver in mcq_solve_kernel).
__device__ __noinline__ int ipm_box(const LCtx& c, double tol, bool resume)
{
    // Pointers and the per-thread index set are re-derived at the top of every pass: nothing but a few scalars is live
    // across the (non-inlined) factorisation / solve calls, so nothing is spilled to scratch and reloaded around them.
#define IPB_SETUP                                                                                                      \
    const int tid = threadIdx.x, n = c.d.n, nm = c.nm;                                                                 \
    double* red = g_sm + SM_RED;                                                                                       \
    const gdouble* LO = VEC(c.w, nm, V_LO);                                                                            \
    const gdouble* HI = VEC(c.w, nm, V_HI);                                                                            \
    gdouble* X = VEC(c.w, nm, V_X);                                                                                    \
    gdouble* G = VEC(c.w, nm, V_G);                                                                                    \
    gdouble* ZL = VEC(c.w, nm, V_ZL);                                                                                  \
    gdouble* ZU = VEC(c.w, nm, V_ZU);                                                                                  \
    gdouble* SIG = VEC(c.w, nm, V_SIG);                                                                                \
    gdouble* RHS = VEC(c.w, nm, V_RHS);                                                                                \
    gdouble* DXA = VEC(c.w, nm, V_DXA);                                                                                \
    gschar* ST = c.w.state;                                                                                            \
    gdouble* IND = VEC(c.w, nm, V_T3);                                                                                 \
    (void)IND; (void)red; (void)LO; (void)HI; (void)X; (void)G; (void)ZL; (void)ZU; (void)SIG; (void)RHS; (void)DXA; (void)ST;    \
    int idx[IPB_E];                                                                                                    \
    bool ok[IPB_E];                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < IPB_E; ++u) {                                                                \
        ok[u] = tid + u * MCQ_NT < n;                                                                                  \
        idx[u] = ok[u] ? tid + u * MCQ_NT : 0;          /* entry 0 stands in for the absent ones (loaded, never stored) */ \
    }
    // the same for one half (entries h * IPB_H .. + IPB_H) of a pass: idx / ok of IPB_H entries
#define IPB_HALF(h)                                                                                                    \
    int idx[IPB_H];                                                                                                    \
    bool ok[IPB_H];                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < IPB_H; ++u) {                                                                \
        ok[u] = tid + ((h) * IPB_H + u) * MCQ_NT < n;                                                                  \
        idx[u] = ok[u] ? tid + ((h) * IPB_H + u) * MCQ_NT : 0;                                                         \
    }
#define IPB_PTRS                                                                                                       \
    const int tid = threadIdx.x, n = c.d.n, nm = c.nm;                                                                 \
    double* red = g_sm + SM_RED;                                                                                       \
    const gdouble* LO = VEC(c.w, nm, V_LO);                                                                            \
    const gdouble* HI = VEC(c.w, nm, V_HI);                                                                            \
    gdouble* X = VEC(c.w, nm, V_X);                                                                                    \
    gdouble* G = VEC(c.w, nm, V_G);                                                                                    \
    gdouble* ZL = VEC(c.w, nm, V_ZL);                                                                                  \
    gdouble* ZU = VEC(c.w, nm, V_ZU);                                                                                  \
    gdouble* SIG = VEC(c.w, nm, V_SIG);                                                                                \
    gdouble* RHS = VEC(c.w, nm, V_RHS);                                                                                \
    gdouble* DXA = VEC(c.w, nm, V_DXA);                                                                                \
    gschar* ST = c.w.state;                                                                                            \
    gdouble* IND = VEC(c.w, nm, V_T3);                                                                                 \
    (void)IND; (void)red; (void)LO; (void)HI; (void)X; (void)G; (void)ZL; (void)ZU; (void)SIG; (void)RHS; (void)DXA; (void)ST;
    const double zscale = c.zscale;
    const double IPM_TOL = tol;
    c.out_iters = 0;
    {
    IPB_SETUP
#pragma unroll
    for (int u = 0; u < IPB_E; ++u) {
        if (!ok[u]) continue;
        const int i = idx[u];
        const bool fixed = !(HI[i] - LO[i] > 1e-12);
        ST[i] = fixed ? 2 : 0;
        if (!resume) {
            X[i] = 0.5 * (LO[i] + HI[i]);
            ZL[i] = fixed ? 0.0 : zscale;
            ZU[i] = ZL[i];
        }
    }
    }
    __syncthreads();
    if (resume) gradient(c, VEC(c.w, c.nm, V_X), nullptr, VEC(c.w, c.nm, V_T0), VEC(c.w, c.nm, V_G));
    const double npairs = 2.0 * c.nfree;
    const bool any_fixed = c.nfree < (double)c.d.n;
    if (!(npairs > 0.0)) return MCQ_OK;

    // g = H x + f is carried along (see ipm()): exact on entry, exact again before convergence is declared
    bool g_exact = true;
    double mu_prev = 1e300;
    int stalled = 0;
    // Mixed precision (round 4): the FIRST factorisations of the cold interior point store their records as floats (mcq_kkt.inc, KRec).
    // A direction from such records is exact to ~6e-8 of its own size: an inexact Newton step, which the iteration absorbs -- but on a row
    // close to its bound, sig dx is of the size of the multiplier, so the residual of the solve is ~6e-8 z there: that is how much the carried
    // gradient drifts per iteration, and as low as the dual residual can get on such records.  So they serve while the dual residual is
    // far above that (MCQ_IPM_F32_RD, relative to the gradient scale: the first four or five of ~11 iterations); at the switch the gradient
    // is recomputed exactly once, and the rest of the phase runs on fp64 records as before.
#ifndef MCQ_IPM_F32
#define MCQ_IPM_F32 1
#endif
#ifndef MCQ_IPM_F32_RD
#define MCQ_IPM_F32_RD 1e-5
#endif
    bool f32 = MCQ_IPM_F32 && !resume;
    // The carried gradient is as good as a recomputed one while every step it has absorbed since the last exact computation came from fp64
    // records (measured: 6e-15 .. 9e-15 of the gradient scale after the fp64 iterations of a run, against a tolerance of 1e-10; a step from
    // float records leaves 1e-7): convergence is then declared without the confirming E'(E x) -- one gradient per problem less.  Not in the
    // resumed attempt (tolerance 1e-13).  (-DMCQ_IPM_TRUST_FP64_CARRY=0: always confirm, as rounds 2-4 did.)
#ifndef MCQ_IPM_TRUST_FP64_CARRY
#define MCQ_IPM_TRUST_FP64_CARRY 1
#endif
    bool g_f32 = false;
    for (int it = 1; it <= c.max_ipm_iter; ++it) {
        // ---- pass 1: complementarity, dual residual, sig, predictor right-hand side ----------------------------------------
        double mu;
        for (;;) {
            IPB_PTRS
            double rdm = 0.0;
            mu = 0.0;
            {
#pragma unroll
            for (int h = 0; h < IPB_E / IPB_H; ++h) {
                IPB_HALF(h)
                double x[IPB_H], lo[IPB_H], hi[IPB_H], zl[IPB_H], zu[IPB_H], g[IPB_H];
                int st[IPB_H];
#pragma unroll
                for (int u = 0; u < IPB_H; ++u) {
                    const int i = idx[u];
                    st[u] = ST[i]; x[u] = X[i]; lo[u] = LO[i]; hi[u] = HI[i]; zl[u] = ZL[i]; zu[u] = ZU[i]; g[u] = G[i];
                }
#pragma unroll
                for (int u = 0; u < IPB_H; ++u) {
                    if (!ok[u]) continue;
                    const int i = idx[u];
                    if (st[u] != 0) { RHS[i] = 0.0; continue; }
                    const double sl = x[u] - lo[u], su = hi[u] - x[u];
                    mu += sl * zl[u] + su * zu[u];
                    rdm = fmax(rdm, fabs(g[u] - zl[u] + zu[u]));
                    SIG[i] = zl[u] / sl + zu[u] / su;
                    RHS[i] = -g[u];
                }
            }
            block_reduce2_(mu, 0, rdm, 2, red);
            mu /= npairs;
            }
            const bool conv = mu < IPM_TOL * zscale * c.wmean && rdm < IPM_TOL * zscale;
#ifdef IPM_TRACE
            if (threadIdx.x == 0) printf("ipmb it %d mu %.3e rdm %.3e exact %d resume %d\n", it, mu / (zscale * c.wmean), rdm / zscale, (int)g_exact, (int)resume);
#endif
            if (f32 && it > 1 && rdm < MCQ_IPM_F32_RD * zscale) {      // float records have done their part
                f32 = false;
                if (!g_exact) {
                    gradient(c, X, nullptr, VEC(c.w, nm, V_T0), G);
                    g_exact = true;
                    g_f32 = false;
                    continue;
                }
            }
            if (conv && (g_exact || (MCQ_IPM_TRUST_FP64_CARRY && !resume && !g_f32))) return MCQ_OK;
            if (!conv) break;
            gradient(c, X, nullptr, VEC(c.w, nm, V_T0), G);     // looks converged on the carried gradient: confirm on the exact one
            g_exact = true;
            g_f32 = false;
        }
        c.out_iters = it;
        if (resume) {
            // at the tighter tolerance round-off can stop the complementarity from shrinking: three rounds without a 10 %
            // reduction end the attempt (the active-set phase then decides)
            stalled = mu > 0.9 * mu_prev ? stalled + 1 : 0;
            mu_prev = mu;
            if (stalled >= 3) return MCQ_OK;
        }

        // ---- factorisation, predictor solve -----------------------------------------------------------------------------
        // the predictor's right-hand side (written in pass 1) rides through the factorisation: its forward substitution is done
        // when the factor is
        const int fs = timed_factor(c, VEC(c.w, c.nm, V_SIG), any_fixed ? c.w.state : nullptr,
                                    VEC(c.w, c.nm, V_RHS), f32);
        // resumed attempt (complementarity already below 1e-10): an iterate that sits ON a bound in floating point (slack 0, sig = inf) ends
        // the attempt like a stalled complementarity does -- the pairs of the last completed iteration go to the active-set phase
        if (fs != 0) return (resume && fs == MCQ_NOT_PD) ? MCQ_OK : fs;
        timed_solve(c, VEC(c.w, c.nm, V_RHS), true);

        // ---- pass 2: affine step lengths, centring parameter, corrector right-hand side: ONE load phase (six arrays of eight entries stay in
        //      registers across the two block reductions; the affine multiplier steps are recomputed where they are needed) ----------
        double smu;
        {
            IPB_SETUP
            double dxa[IPB_E], sl[IPB_E], su[IPB_E], zl[IPB_E], zu[IPB_E], g[IPB_E];
            bool act[IPB_E];
#pragma unroll
            for (int u = 0; u < IPB_E; ++u) {
                const int i = idx[u];
                const double x = X[i];
                act[u] = ok[u] && ST[i] == 0;
                dxa[u] = RHS[i]; sl[u] = x - LO[i]; su[u] = HI[i] - x; zl[u] = ZL[i]; zu[u] = ZU[i]; g[u] = G[i];
            }
            double ap = 1.0, ad = 1.0;
#pragma unroll
            for (int u = 0; u < IPB_E; ++u) {
                if (!ok[u]) continue;
                if (!act[u]) { DXA[idx[u]] = 0.0; continue; }
                const double dx = dxa[u];
                DXA[idx[u]] = dx;
                const double dzla = -zl[u] - zl[u] * dx / sl[u];
                const double dzua = -zu[u] + zu[u] * dx / su[u];
                if (dx < 0.0) ap = fmin(ap, -sl[u] / dx);
                if (dx > 0.0) ap = fmin(ap, su[u] / dx);
                if (dzla < 0.0) ad = fmin(ad, -zl[u] / dzla);
                if (dzua < 0.0) ad = fmin(ad, -zu[u] / dzua);
            }
            block_reduce2_(ap, 1, ad, 1, red);
            double mua = 0.0;
#pragma unroll
            for (int u = 0; u < IPB_E; ++u) {
                if (!act[u]) continue;
                const double dzla = -zl[u] - zl[u] * dxa[u] / sl[u];
                const double dzua = -zu[u] + zu[u] * dxa[u] / su[u];
                mua += (sl[u] + ap * dxa[u]) * (zl[u] + ad * dzla) + (su[u] - ap * dxa[u]) * (zu[u] + ad * dzua);
            }
            mua = block_reduce_(mua, 0, red) / npairs;
            const double ratio = mua / mu;
            smu = ratio * ratio * ratio * mu;
#pragma unroll
            for (int u = 0; u < IPB_E; ++u) {
                if (!ok[u]) continue;
                const double dzla = -zl[u] - zl[u] * dxa[u] / sl[u];
                const double dzua = -zu[u] + zu[u] * dxa[u] / su[u];
                RHS[idx[u]] = act[u] ? -g[u] + (smu - dxa[u] * dzla) / sl[u] - (smu + dxa[u] * dzua) / su[u] : 0.0;
            }
        }
        timed_solve(c, VEC(c.w, c.nm, V_RHS));

        // ---- pass 3: step length of the combined direction, update: ONE load phase -- nine arrays of eight entries stay in registers
        //      across the block reduction; the multiplier steps and H dx are recomputed after it instead of being kept (at 256 VGPRs --
        //      two workgroups per CU -- twelve arrays spilled).  (Folding the next iteration's residual pass into this one was tried: the
        //      extra live values cost more than the seven vector reads it saves.) --------------------------------------------------
        {
            IPB_SETUP
            // fraction of the way to the boundary: 0.995 far from the solution, closer to 1 as the complementarity shrinks
            // (Mehrotra's adaptive rule; saves a third of an iteration on average, scripts/proto_ipm.py "adaptive step")
#ifdef MCQ_IPM_FIXED_STEP
            const double gm = 0.995;
#else
            const double gm = fmin(fmax(MCQ_IPM_GM_MIN, 1.0 - MCQ_IPM_GM_C * mu / (zscale * c.wmean)), 1.0 - 1e-9);
#endif
            double amax = 1.0 / gm;
            double dx[IPB_E], da[IPB_E], x[IPB_E], lo[IPB_E], hi[IPB_E], zl[IPB_E], zu[IPB_E], g[IPB_E], sg[IPB_E];
            bool act[IPB_E];
#pragma unroll
            for (int u = 0; u < IPB_E; ++u) {
                const int i = idx[u];
                act[u] = ok[u] && ST[i] == 0;
                dx[u] = RHS[i]; da[u] = DXA[i]; x[u] = X[i]; lo[u] = LO[i]; hi[u] = HI[i]; zl[u] = ZL[i]; zu[u] = ZU[i];
                g[u] = G[i]; sg[u] = SIG[i];
            }
#define IPB_STEP3(u)                                                                                                          \
                const double sl = x[u] - lo[u], su = hi[u] - x[u];                                                              \
                const double dzla = -zl[u] - zl[u] * da[u] / sl, dzua = -zu[u] + zu[u] * da[u] / su;                             \
                const double dzl = (-sl * zl[u] + smu - da[u] * dzla - zl[u] * dx[u]) / sl;                                     \
                const double dzu = (-su * zu[u] + smu + da[u] * dzua + zu[u] * dx[u]) / su;
#pragma unroll
            for (int u = 0; u < IPB_E; ++u) {
                if (!act[u]) continue;
                IPB_STEP3(u)
                if (dx[u] < 0.0) amax = fmin(amax, -sl / dx[u]);
                if (dx[u] > 0.0) amax = fmin(amax, su / dx[u]);
                if (dzl < 0.0) amax = fmin(amax, -zl[u] / dzl);
                if (dzu < 0.0) amax = fmin(amax, -zu[u] / dzu);
            }
            amax = block_reduce_(amax, 1, red);
            const double a = fmin(1.0, gm * amax);
            c.last_step = a;
#pragma unroll
            for (int u = 0; u < IPB_E; ++u) {
                if (!act[u]) continue;
                const int i = idx[u];
                IPB_STEP3(u)
                // H dx = (corrector right-hand side) - sig dx
                const double rc = -g[u] + (smu - da[u] * dzla) / sl - (smu + da[u] * dzua) / su;
                const double hdx = rc - sg[u] * dx[u];
                const double gn = g[u] + a * hdx, xn = x[u] + a * dx[u], zln = zl[u] + a * dzl, zun = zu[u] + a * dzu;
                G[i] = gn;
                X[i] = xn;
                ZL[i] = zln;
                ZU[i] = zun;
                // Tapia indicators of this step for the active-set identification (the last step's survive)
                const double rsl = (sl + a * dx[u]) * zl[u], rzl = (zl[u] + a * dzl) * sl;     // s+/s < z+/z  <=>  s+ z < z+ s
                const double rsu = (su - a * dx[u]) * zu[u], rzu = (zu[u] + a * dzu) * su;
                const bool al = rsl < MCQ_TAPIA_RATIO * rzl && sl + a * dx[u] < MCQ_TAPIA_SHRINK * sl, au = rsu < MCQ_TAPIA_RATIO * rzu && su - a * dx[u] < MCQ_TAPIA_SHRINK * su;
                IND[i] = al ? -1.0 : (au ? 1.0 : 0.0);
            }
#undef IPB_STEP3
        }
        g_exact = false;
        g_f32 |= f32;
        __syncthreads();
    }
    return MCQ_ITER_CAP;
#undef IPB_SETUP
}

// ---------------------------------------------------------------------------------------------------------------------
// Active-set identification from the interior-point pairs + block principal pivoting (Kim-Park / Judice-Pires rule
// with single-pivot backup) on the vertex: every iterate solves the equality-constrained problem of its working set
// exactly, so the returned point is an exact KKT vertex like the one a dual active-set (Goldfarb-Idnani) solver returns.
// Working set = pinned box rows (masked banded Cholesky of H_FF) + at most MCQ_KMAX active curvature rows, the latter
// through the Schur complement  S = E_K M^-1 E_K'  (|K| extra banded solves per iteration; rare path).
// ---------------------------------------------------------------------------------------------------------------------
// ---- curvature rows of the working set: the Schur complement  S = E_K M^-1 E_K'  (nk x nk) --------------------------------------
// Where the working set's arrays live (KappaMem).  Up to MCQ_KMAX rows -- every case the reference's tracks produce -- the index /
// sign / pivot lists and the three nk-vectors sit in LDS, S in HBM behind the scratch vector of McqWork.Z (row-major, leading
// dimension MCQ_KMAX) and its LU factorisation with partial pivoting runs on a copy in the LDS overlay, all 256 threads on the
// rank-1 updates (S is SPD for independent rows; pivoting keeps a nearly dependent working set from blowing up).  Beyond that
// (round 3: quadprog has no such limit) the problem claims one of the handle's overflow slots (McqBatch::kbig, MCQ_KBIG rows): every
// array in HBM, the elimination in place in HBM -- slow, rare, exact.
struct KappaMem {
    double* kmu;     // multipliers / solution of the Schur system
    double* krh;     // its right-hand side
    double* mul;     // elimination multipliers of the current column
    int* ki;         // ki[0] = nk, ki[1 + q] = row, ki[1 + cap + q] = sign
    int* piv;        // pivot rows
    gdouble* sg;     // S / its factors, row-major
    int cap, ld;     // rows the arrays hold, leading dimension of sg
    bool in_lds;     // S is factored on a copy in the LDS overlay
};
__device__ __forceinline__ KappaMem kappa_mem_lds(const LCtx& c)
{
    KappaMem k;
    k.kmu = g_sm + SM_KV;
    k.krh = g_sm + SM_KV + MCQ_KMAX;
    k.mul = g_sm + SM_KV + 2 * MCQ_KMAX;
    k.ki = (int*)(g_sm + SM_KI);
    k.piv = (int*)(g_sm + SM_KI) + 1 + 2 * MCQ_KMAX;
    k.sg = c.w.Z + c.nm;
    k.cap = MCQ_KMAX;
    k.ld = MCQ_KMAX;
    k.in_lds = true;
    return k;
}
__device__ __forceinline__ KappaMem kappa_mem_slot(double* slot)
{
    KappaMem k;
    k.sg = (gdouble*)slot;
    double* v = slot + (size_t)MCQ_KBIG * MCQ_KBIG;
    k.kmu = v;
    k.krh = v + MCQ_KBIG;
    k.mul = v + 2 * MCQ_KBIG;
    k.ki = (int*)(v + 3 * MCQ_KBIG);                 // 1 + 2 KBIG ints
    k.piv = k.ki + 2 + 2 * MCQ_KBIG;                // KBIG ints
    k.cap = MCQ_KBIG;
    k.ld = MCQ_KBIG;
    k.in_lds = false;
    return k;
}

// LU with partial pivoting of S (nk x nk, leading dimension ld) in place; A: where the elimination runs (the LDS copy, leading
// dimension nk, or S itself)
template <typename PA>
__device__ __forceinline__ void kappa_lu_eliminate(PA A, int lda, int nk, double* mul, int* piv)
{
    const int tid = threadIdx.x;
    double* red = g_sm + SM_RED;
    for (int cidx = 0; cidx < nk; ++cidx) {
        // pivot row: largest magnitude of the column below the diagonal (block-wide argmax; ties to the lowest row, as a serial scan)
        double best = -1.0;
        int brow = cidx;
        for (int r = cidx + tid; r < nk; r += MCQ_NT) {
            const double v = fabs(A[(size_t)r * lda + cidx]);
            if (v > best) { best = v; brow = r; }
        }
        const double bmax = block_reduce_(best, 2, red);
        double cand = (best == bmax && best >= 0.0) ? (double)brow : 1e300;
        cand = block_reduce_(cand, 1, red);
        const int pr = cand < 1e299 ? (int)cand : cidx;
        if (tid == 0) piv[cidx] = pr;
        if (pr != cidx)
            for (int cc = tid; cc < nk; cc += MCQ_NT) {
                const double t = A[(size_t)cidx * lda + cc];
                A[(size_t)cidx * lda + cc] = A[(size_t)pr * lda + cc];
                A[(size_t)pr * lda + cc] = t;
            }
        __syncthreads();
        const double pv = A[(size_t)cidx * lda + cidx];
        for (int r = cidx + 1 + tid; r < nk; r += MCQ_NT) mul[r] = pv != 0.0 ? A[(size_t)r * lda + cidx] / pv : 0.0;
        __syncthreads();
        const int rem = nk - cidx - 1;
        for (int e = tid; e < rem * rem; e += MCQ_NT) {
            const int r = cidx + 1 + e / rem, cc = cidx + 1 + e % rem;
            A[(size_t)r * lda + cc] -= mul[r] * A[(size_t)cidx * lda + cc];
        }
        __syncthreads();
        for (int r = cidx + 1 + tid; r < nk; r += MCQ_NT) A[(size_t)r * lda + cidx] = mul[r];      // L below the diagonal
        __syncthreads();
    }
}

__device__ void kappa_lu_factor(const KappaMem& K, int nk)
{
    const int tid = threadIdx.x;
    __syncthreads();
    if (K.in_lds && nk * nk <= OVL_SIZE) {          // (the saddle-point core's overlay holds 86 x 86: beyond that the elimination runs in HBM)
        double* A = g_sm + SM_OVL;                       // nk x nk, leading dimension nk
        for (int e = tid; e < nk * nk; e += MCQ_NT) A[e] = K.sg[(size_t)(e / nk) * K.ld + (e % nk)];
        __syncthreads();
        kappa_lu_eliminate(A, nk, nk, K.mul, K.piv);
        for (int e = tid; e < nk * nk; e += MCQ_NT) K.sg[(size_t)(e / nk) * K.ld + (e % nk)] = A[e];
    } else {
        kappa_lu_eliminate(K.sg, K.ld, nk, K.mul, K.piv);
    }
    __syncthreads();
}

// kmu <- S^-1 krh with the factors kappa_lu_factor left in sg (row interchanges applied to the right-hand side first)
template <typename PA>
__device__ __forceinline__ void kappa_lu_substitute(PA A, int lda, int nk, double* KMU, const int* piv)
{
    const int tid = threadIdx.x;
    if (tid == 0)
        for (int cidx = 0; cidx < nk; ++cidx) {
            const int pr = piv[cidx];
            if (pr != cidx) { const double t = KMU[cidx]; KMU[cidx] = KMU[pr]; KMU[pr] = t; }
        }
    __syncthreads();
    for (int cidx = 0; cidx < nk; ++cidx) {               // L y = P b   (unit lower triangle)
        const double yc = KMU[cidx];
        __syncthreads();
        for (int r = cidx + 1 + tid; r < nk; r += MCQ_NT) KMU[r] -= A[(size_t)r * lda + cidx] * yc;
        __syncthreads();
    }
    for (int cidx = nk - 1; cidx >= 0; --cidx) {          // U x = y
        if (tid == 0) { const double pv = A[(size_t)cidx * lda + cidx]; KMU[cidx] = pv != 0.0 ? KMU[cidx] / pv : 0.0; }
        __syncthreads();
        const double xc = KMU[cidx];
        for (int r = tid; r < cidx; r += MCQ_NT) KMU[r] -= A[(size_t)r * lda + cidx] * xc;
        __syncthreads();
    }
}

__device__ void kappa_lu_solve(const KappaMem& K, int nk)
{
    const int tid = threadIdx.x;
    __syncthreads();
    for (int q = tid; q < nk; q += MCQ_NT) K.kmu[q] = K.krh[q];
    if (K.in_lds && nk * nk <= OVL_SIZE) {
        double* A = g_sm + SM_OVL;
        for (int e = tid; e < nk * nk; e += MCQ_NT) A[e] = K.sg[(size_t)(e / nk) * K.ld + (e % nk)];
        __syncthreads();
        kappa_lu_substitute((const double*)A, nk, nk, K.kmu, K.piv);
    } else {
        __syncthreads();
        kappa_lu_substitute((const gdouble*)K.sg, K.ld, nk, K.kmu, K.piv);
    }
}

// v <- M^-1 (E_K' KMU restricted to the free set):  the multipliers scattered onto their rows (Q), one band product, one solve
__device__ void kappa_apply(const LCtx& c, const KappaMem& K, int nk, gdouble* Q, gdouble* v)
{
    const int tid = threadIdx.x, n = c.d.n;
    const double* KMU = K.kmu;
    const int* KI = K.ki;
    const gschar* ST = c.w.state;
    for (int i = tid; i < n; i += MCQ_NT) Q[i] = 0.0;
    __syncthreads();
    for (int q = tid; q < nk; q += MCQ_NT) Q[KI[1 + q]] = KMU[q];
    __syncthreads();
    apply_Et(c, Q, v);
    __syncthreads();
    for (int i = tid; i < n; i += MCQ_NT) if (ST[i] != 0) v[i] = 0.0;
    __syncthreads();
    timed_solve(c, v);
}

// identify: 1 the working set from the interior point's pairs; 0 the box rows given (state), the curvature rows from the pairs;
//           2 both given (state, V_SK): the vertex the Goldfarb-Idnani path arrived at
__device__ __noinline__ int active_set(const LCtx& c, bool with_kappa, bool tapia, int cap, int identify, const KappaMem K)
{
    const int tid = threadIdx.x, n = c.d.n, nm = c.nm;
    double* red = g_sm + SM_RED;
    double* KMU = K.kmu;
    double* KRH = K.krh;
    int* KI = K.ki;                        // KI[0] = nk, KI[1+q] = row, KI[1+cap+q] = sign
    const int kcap = K.cap;
    const gdouble* LO = VEC(c.w, nm, V_LO);
    const gdouble* HI = VEC(c.w, nm, V_HI);
    const gdouble* KR = VEC(c.w, nm, V_KREF);
    gdouble* X = VEC(c.w, nm, V_X);
    gdouble* G = VEC(c.w, nm, V_G);
    gdouble* ZL = VEC(c.w, nm, V_ZL);
    gdouble* ZU = VEC(c.w, nm, V_ZU);
    gdouble* RHS = VEC(c.w, nm, V_RHS);
    gdouble* T0 = VEC(c.w, nm, V_T0);
    gdouble* T1 = VEC(c.w, nm, V_T1);
    gdouble* T2 = VEC(c.w, nm, V_T2);
    gdouble* T3 = VEC(c.w, nm, V_T3);
    gdouble* Q = VEC(c.w, nm, V_Q);
    gdouble* KF = VEC(c.w, nm, V_SK);      // per-row flag of the curvature working set: 0, +1 (upper), -1 (lower)
    gdouble* PV = VEC(c.w, nm, V_DXA);     // by how much a free row leaves its box in this round (0 elsewhere)
    gschar* ST = c.w.state;
    const double zscale = c.zscale, fscale = c.fscale, kb = c.kbound;
    const int win = MCQ_AS_WINDOW < (n - 1) / 2 ? MCQ_AS_WINDOW : (n - 1) / 2;
    c.out_iters = 0;
    c.out_kkt = 0.0;
    c.out_nk = 0;

    // ---- identification ---------------------------------------------------------------------------------------------------
    // (identify == false: the working set is given -- carried over from the previous IQP pass -- and the pairs are not read)
    for (int i = tid; i < n; i += MCQ_NT) {
        if (identify == 1 && ST[i] == 0) {
            // magnitude test on the final pair: active when the scaled multiplier exceeds the scaled slack
            const double wdt = HI[i] - LO[i];
            const double sl = X[i] - LO[i], su = HI[i] - X[i];
            signed char st = 0;
            if (sl * zscale < ZL[i] * wdt) st = -1;
            else if (su * zscale < ZU[i] * wdt) st = 1;
            // ... plus the rows it misses at mu = 1e-10 (weakly active: slack and multiplier both ~ sqrt(mu)) when the last
            // interior-point step gave Tapia evidence for them: slack shrinking faster than the multiplier,
            // s+/s < MCQ_TAPIA_SHRINK and s+/s < MCQ_TAPIA_RATIO z+/z.  One-sided (it only adds rows).  The thresholds were
            // 0.5 / 0.3 while a free row pinned by mistake could send block pivoting into dozens of rounds; with the
            // one-row-per-neighbourhood exchange below a false positive costs one round like a miss does, and 0.7 / 0.7
            // measured best on the 3 x 1024 problems of the IQP workload (rounds 1.60 -> 1.27 in the first passes;
            // 1.0 / 1.0: 1.23 but a 6-round straggler).
            if (tapia && st == 0) st = (signed char)VEC(c.w, nm, V_T3)[i];
            ST[i] = st;
        }
        double kf = 0.0;
        if (with_kappa && identify == 2) kf = KF[i];
        else if (with_kappa) {
            const gdouble* TL = VEC(c.w, nm, V_TL);
            const gdouble* TU = VEC(c.w, nm, V_TU);
            const gdouble* YL = VEC(c.w, nm, V_YL);
            const gdouble* YU = VEC(c.w, nm, V_YU);
            const double mu0 = 0.5 * zscale * c.wmean;
            if (TL[i] * mu0 < YL[i] * kb * kb) kf = -1.0;
            else if (TU[i] * mu0 < YU[i] * kb * kb) kf = 1.0;
        }
        KF[i] = kf;
    }
    __syncthreads();
    const double TOLX = 1e-10;
    const double toly = 1e-10 * (fscale > 0.0 ? fscale : 1.0);
    const double tolk = 1e-10 * kb;
    int best = 2 * n + 1, pcnt = 3;
    for (int it = 1; it <= cap; ++it) {
        c.out_iters = it;
        // ---- compact list of the curvature working set (thread 0; n is small relative to everything else here) ------------
        if (tid == 0) {
            int nk = 0;
            if (with_kappa)
                for (int i = 0; i < n; ++i)
                    if (KF[i] != 0.0 && nk < kcap + 1) {
                        if (nk < kcap) { KI[1 + nk] = i; KI[1 + kcap + nk] = KF[i] > 0.0 ? 1 : -1; }
                        ++nk;
                    }
            KI[0] = nk;
        }
        __syncthreads();
        const int nk = KI[0];
        if (nk > kcap) return MCQ_KAPPA_ACTIVE;          // more active curvature rows than these arrays hold (the caller may retry
                                                         // with an overflow slot)
        c.out_nk = nk;

        for (int i = tid; i < n; i += MCQ_NT) {
            const signed char st = ST[i];
            T1[i] = st == 0 ? 0.0 : (st < 0 ? LO[i] : (st == 1 ? HI[i] : 0.5 * (LO[i] + HI[i])));
        }
        gradient(c, T1, nullptr, T0, T2);       // T2 = H x_A + f
        for (int i = tid; i < n; i += MCQ_NT) RHS[i] = ST[i] == 0 ? -T2[i] : T1[i];
        const int fs = timed_factor(c, nullptr, ST, RHS);
        if (fs != 0) return fs;
        timed_solve(c, RHS, true);   // x0 (pinned rows carry their bounds)
        if (nk > 0) {
            // column q of S = E_K (M^-1 E_kq' restricted to the free set): per active row one application of E' (to a unit vector), one
            // solve, one application of E -- nothing but one scratch vector kept (x = x0 - M^-1 E_K' mu costs one more solve afterwards
            // instead of nk stored columns).  (Rounds 1-3 read rows of a 65-wide band of E here; E is applied through the spline system
            // now, untruncated, and no band exists.)
            gdouble* Zs = c.w.Z;
            gdouble* SG = K.sg;
            for (int q = 0; q < nk; ++q) {
                const int k = KI[1 + q];
                for (int i = tid; i < n; i += MCQ_NT) T2[i] = i == k ? 1.0 : 0.0;
                apply_Et(c, T2, Zs);
                for (int i = tid; i < n; i += MCQ_NT) if (ST[i] != 0) Zs[i] = 0.0;
                __syncthreads();
                timed_solve(c, Zs);
                tri_apply_E(c, Zs, nullptr, 0.0, T0);
                for (int q2 = tid; q2 < nk; q2 += MCQ_NT) SG[(size_t)q2 * K.ld + q] = T0[KI[1 + q2]];
                __syncthreads();
            }
            // rhs = E_K x0 + k_ref - s kb
            tri_apply_E(c, RHS, KR, 1.0, T0);
            for (int q = tid; q < nk; q += MCQ_NT) KRH[q] = T0[KI[1 + q]] - KI[1 + kcap + q] * kb;
            kappa_lu_factor(K, nk);
            kappa_lu_solve(K, nk);
            kappa_apply(c, K, nk, Q, Zs);       // Q = multipliers on their rows, Zs = M^-1 E_K' mu
            for (int i = tid; i < n; i += MCQ_NT) RHS[i] -= Zs[i];
            __syncthreads();
        }
        for (int i = tid; i < n; i += MCQ_NT) X[i] = ST[i] == 0 ? RHS[i] : T1[i];
        gradient(c, X, nk > 0 ? Q : nullptr, T0, G);      // Lagrangian gradient  H x + f + E_K' mu
        if (with_kappa) tri_apply_E(c, X, KR, 1.0, T2);      // r = E x + k_ref
        double nv = 0.0, imax = -1.0, kk = 0.0;
        for (int i = tid; i < n; i += MCQ_NT) {
            const signed char st = ST[i];
            int v = 0;
            double pv = 0.0;
            if (st == 0) {
                if (X[i] < LO[i] - TOLX) { v = -1; pv = LO[i] - X[i]; }
                else if (X[i] > HI[i] + TOLX) { v = 1; pv = X[i] - HI[i]; }
                kk = fmax(kk, fabs(G[i]));
            } else if (st == -1) { if (G[i] < -toly) v = 2; }
            else if (st == 1) { if (G[i] > toly) v = 2; }
            PV[i] = pv;
            int vk = 0;
            if (with_kappa) {
                const double kf = KF[i];
                if (kf == 0.0) {
                    if (T2[i] > kb + tolk) vk = 1;
                    else if (T2[i] < -kb - tolk) vk = -1;
                } else if (kf * Q[i] < -toly) vk = 2;      // multiplier of an active row must push inward
            }
            T3[i] = (double)(v + 8 * vk);
            if (v != 0) { nv += 1.0; imax = fmax(imax, (double)i); }
            if (vk != 0) { nv += 1.0; imax = fmax(imax, (double)(n + i)); }
        }
        nv = block_reduce_(nv, 0, red);
        imax = block_reduce_(imax, 2, red);
        c.out_kkt = block_reduce_(kk, 2, red);
        const int nvi = (int)nv;
        if (nvi == 0) {
            // fp64 residual refinement through E on the final working set (same factor); box-only working sets
            // A round whose correction is already below 1e-8 m is the last one: the next would move alpha by (cond * eps)
            // times that, i.e. far below the 1e-9 m at which the dense oracle itself is known.
            for (int r = 0; r < c.refine_steps; ++r) {
                for (int i = tid; i < n; i += MCQ_NT) RHS[i] = ST[i] == 0 ? -G[i] : 0.0;
                timed_solve(c, RHS);
                if (nk > 0) {
                    // curvature rows in the working set: one step on the KKT system [M E_K'; E_K 0] through the factored Schur
                    // complement -- dx0 = M^-1(-g), S dmu = E_K dx0 + (E_K x + k_ref - s kb), dx = dx0 - M^-1 E_K' dmu
                    gdouble* Zs = c.w.Z;
                    tri_apply_E(c, RHS, nullptr, 0.0, T0);
                    for (int q = tid; q < nk; q += MCQ_NT) {
                        const int k = KI[1 + q];
                        KRH[q] = T0[k] + T2[k] - KI[1 + kcap + q] * kb;
                    }
                    kappa_lu_solve(K, nk);
                    for (int q = tid; q < nk; q += MCQ_NT) KRH[q] = KMU[q];       // dmu (kappa_apply reads KMU, Q is rebuilt below)
                    __syncthreads();
                    kappa_apply(c, K, nk, T1, Zs);
                    for (int i = tid; i < n; i += MCQ_NT) RHS[i] -= Zs[i];
                    __syncthreads();
                    for (int q = tid; q < nk; q += MCQ_NT) Q[KI[1 + q]] += KRH[q];
                    __syncthreads();
                }
                double dm = 0.0;
                for (int i = tid; i < n; i += MCQ_NT) if (ST[i] == 0) { X[i] += RHS[i]; dm = fmax(dm, fabs(RHS[i])); }
                dm = block_reduce_(dm, 2, red);
                gradient(c, X, nk > 0 ? Q : nullptr, T0, G);
                if (nk > 0) tri_apply_E(c, X, KR, 1.0, T2);      // r = E x + k_ref
                c.refine_rounds = r + 1;
                if (!(dm > 1e-8)) break;
            }
            double k2 = 0.0;
            for (int i = tid; i < n; i += MCQ_NT) if (ST[i] == 0) k2 = fmax(k2, fabs(G[i]));
            c.out_kkt = block_reduce_(k2, 2, red);
            return MCQ_OK;
        }
        bool full;
        if (nvi < best) { best = nvi; pcnt = 3; full = true; }
        else if (pcnt > 0) { --pcnt; full = true; }
        else if (with_kappa) return MCQ_ITER_CAP;
        else full = false;
        // (round 6, tried: a warm-started exchange that is down to <= 3 / 6 / 12 violated rows after five rounds goes row by row from there -- the
        //  third IQP pass of the bench workload then falls back to the cold path 41 / 178 / 328 times instead of 15: full exchanges with the
        //  one-row-per-neighbourhood rule are what settles these, single pivots wander longer.  docs/NOTEBOOK.md R6.5)
        // (The single-pivot backup rule terminates on box rows alone -- the linear complementarity problem of a positive definite Hessian
        //  with simple bounds -- and is kept for them.  With curvature rows in the exchange nothing promises that, and a round costs one
        //  solve per curvature row: of the 220 curvature-tight problems of tests/golden/kappa_tight_fuzz.npz every exchange that ends
        //  does so within 10 rounds without ever reaching the rule, and the three that reach it cycle at ~80 rows until the cap; the
        //  720-point stadium (270 rows) did come back from it, after 19 more rounds -- no cheaper than the 370 steps of the
        //  Goldfarb-Idnani path, which takes over from here instead [docs/NOTEBOOK.md R5.6].)
        for (int i = tid; i < n; i += MCQ_NT) {
            const int code = (int)T3[i];
            // decode v in {-1,0,1,2}, vk in {-1,0,1,2}:  code = v + 8 vk
            int vk = (code + 20) / 8 - 2;
            int v = code - 8 * vk;
            if (v > 2) { v -= 8; vk += 1; }
            if (v != 0 && (full || i == (int)imax)) {
                // A missing active row lets the minimiser leave the box over a whole stretch of neighbours (the Hessian is a
                // fourth-difference operator: the raceline bulges through the wall); pinning the stretch pins rows that are
                // free at the optimum, their multipliers come out with the wrong sign, and on these Hessians the exchange
                // then wanders for dozens of rounds.  Only the row that leaves the box furthest within MCQ_AS_WINDOW rows
                // either side is pinned: the wall touches the raceline where it bulges most.  (Wrong-signed multipliers are
                // all released, and the single-pivot backup rule below still bounds the number of rounds.)
                bool take = true;
                if (full && v != 2) {
                    const double me = PV[i];
                    for (int o = 1; o <= win && take; ++o)
                        if (PV[cyc1(i - o, n)] >= me || PV[cyc1(i + o, n)] > me) take = false;
                }
                if (take) ST[i] = v == 2 ? 0 : (signed char)v;
            }
            if (vk != 0 && (full || n + i == (int)imax)) KF[i] = vk == 2 ? 0.0 : (double)vk;
        }
        __syncthreads();
    }
    return MCQ_ITER_CAP;
}

// ---- slots of the handle (overflow slots of the curvature-row working set, slots of the Goldfarb-Idnani path): flags[s] 0 = free, 1 = taken.
//      Claimed by compare-and-swap, released by the holder: zero whenever no launch is in flight.  wait: spin until one is free -- the holders
//      are resident workgroups that wait for nobody, so they finish.  Uniform over the block; returns the slot or -1. ----
__device__ int claim_slot(int* flags, int nslots, bool wait)
{
    int* sh = (int*)(g_sm + SM_RED);
    __syncthreads();
    if (threadIdx.x == 0) {
        int got = -1;
        for (;;) {
            const int s0 = (int)(blockIdx.x % (unsigned)nslots);       // (workgroups start their scans at different flags)
            for (int k = 0; k < nslots && got < 0; ++k) {
                const int s = s0 + k < nslots ? s0 + k : s0 + k - nslots;
                if (atomicCAS(flags + s, 0, 1) == 0) got = s;
            }
            if (got >= 0 || !wait) break;
            // back off: the holders work for milliseconds, and hundreds of waiting workgroups hammering the flags' cache line slow THEM down
            // (round 5: a launch with 455 problems for 8 slots took 25 s; ~0.2 ms between two scans)
            for (int r = 0; r < 64; ++r) __builtin_amdgcn_s_sleep(127);
        }
        sh[0] = got;
    }
    __syncthreads();
    const int slot = sh[0];
    __syncthreads();
    return slot;
}
__device__ void release_slot(int* flags, int slot)
{
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); atomicExch(flags + slot, 0); }
}

// ---- scalars of a problem: the scales of the tolerances, the working set and the iterate at the box centre, the gradient there ----
#ifdef MCQ_OUTLINE_PROLOGUE     /* A/B switch: as calls they cost the solver kernel 64 bytes of scratch per lane and ~0.5 % (round 5) */
#define MCQ_FN_PROLOGUE __device__ __noinline__
#else
#define MCQ_FN_PROLOGUE __device__ __forceinline__
#endif
MCQ_FN_PROLOGUE void problem_scales(const LCtx& c)
{
    const int tid = threadIdx.x, n = c.d.n, nm = c.nm;
    double* red = g_sm + SM_RED;
    const gdouble* LO = VEC(c.w, nm, V_LO);
    const gdouble* HI = VEC(c.w, nm, V_HI);
    const gdouble* F = VEC(c.w, nm, V_F);
    gdouble* X = VEC(c.w, nm, V_X);
    gdouble* G = VEC(c.w, nm, V_G);
    gschar* ST = c.w.state;
    double wsum = 0.0, nfree_d = 0.0, fmaxl = 0.0;
    for (int i = tid; i < n; i += MCQ_NT) {
        const double wdt = HI[i] - LO[i];
        const bool fixed = !(wdt > 1e-12);
        ST[i] = fixed ? 2 : 0;
        X[i] = 0.5 * (LO[i] + HI[i]);
        if (!fixed) { wsum += wdt; nfree_d += 1.0; }
        fmaxl = fmax(fmaxl, fabs(F[i]));
    }
    wsum = block_reduce_(wsum, 0, red);
    const double nfree_s = block_reduce_(nfree_d, 0, red), fscale_s = block_reduce_(fmaxl, 2, red);
    gradient(c, X, nullptr, VEC(c.w, nm, V_T0), G);
    double gm = 0.0;
    for (int i = tid; i < n; i += MCQ_NT) if (ST[i] == 0) gm = fmax(gm, fabs(G[i]));
    double zscale_s = block_reduce_(gm, 2, red);
    if (!(zscale_s > 0.0)) zscale_s = fscale_s > 0.0 ? fscale_s : 1.0;
    if (tid == 0) {
        g_ctx.nfree = nfree_s;
        g_ctx.fscale = fscale_s;
        g_ctx.wmean = nfree_s > 0.0 ? wsum / nfree_s : 1.0;
        g_ctx.zscale = zscale_s;
    }
    __syncthreads();
}

// ---- outputs of a problem: alpha, opt_min_curv's curvature-error post-check, status, diagnostics.  X holds the
//      solution (inside its box), T0 = k_ref + E x and -- dd_valid -- V_TL / V_TU the second derivatives D (n_x x), D (n_y x). ----
struct McqOutcome {
    int status, ipm_iters, as_iters, nact_kappa, gi_iters;
    double kkt, km;
    bool dd_valid;
    long long t_kernel0, c_kernel0, t_epi0;
};
MCQ_FN_PROLOGUE void write_outputs(const LCtx& c, const McqOutcome& r)
{
    const int tid = threadIdx.x, n = c.d.n, nm = c.nm;
    double* red = g_sm + SM_RED;
    const gdouble* X = VEC(c.w, nm, V_X);
    gdouble* T0 = VEC(c.w, nm, V_T0);
    gdouble* T1 = VEC(c.w, nm, V_T1);
    gdouble* T2 = VEC(c.w, nm, V_T2);
    gdouble* T3 = VEC(c.w, nm, V_T3);
    gdouble* Q = VEC(c.w, nm, V_Q);
    const gschar* ST = c.w.state;
    double nact = 0.0;
    for (int i = tid; i < n; i += MCQ_NT) {
        c.w.alpha[i] = X[i];
        if (ST[i] == -1 || ST[i] == 1) nact += 1.0;
    }
    nact = block_reduce_(nact, 0, red);
    const gdouble* XP = VEC(c.w, nm, V_XP);
    const gdouble* YP = VEC(c.w, nm, V_YP);
    const gdouble* XPP = VEC(c.w, nm, V_XPP);
    const gdouble* YPP = VEC(c.w, nm, V_YPP);
    if (!c.direct) {
        for (int i = tid; i < n; i += MCQ_NT) { T1[i] = VEC(c.w, nm, V_NX)[i] * X[i]; T2[i] = VEC(c.w, nm, V_NY)[i] * X[i]; }
        __syncthreads();
        if (r.dd_valid) {
            const gdouble* D1 = VEC(c.w, nm, V_TL);
            const gdouble* D2 = VEC(c.w, nm, V_TU);
            for (int i = tid; i < n; i += MCQ_NT) { T0[i] = D1[i]; T3[i] = D2[i]; }
        } else {
            tri_apply_E(c, X, nullptr, 0.0, Q, T0, T3);      // D (n_x alpha), D (n_y alpha) are by-products of E alpha
        }
        __syncthreads();
    }
    double em = 0.0;
    for (int i = tid; i < (c.direct ? 0 : n); i += MCQ_NT) {
        const int ip = cyc(i + 1, n);
        const double s = VEC(c.w, nm, V_SC)[i];
        const double s2 = s * s;
        const double xpt = XP[i] + (T1[ip] - T1[i]) - (T0[i] + 0.5 * s2 * T0[ip]) / 3.0;
        const double ypt = YP[i] + (T2[ip] - T2[i]) - (T3[i] + 0.5 * s2 * T3[ip]) / 3.0;
        const double xpp = XPP[i] + T0[i], ypp = YPP[i] + T3[i];
        const double xp = XP[i], yp = YP[i];
        const double k0 = (xp * ypp - yp * xpp) / pow(xp * xp + yp * yp, 1.5);
        const double k1 = (xpt * ypp - ypt * xpp) / pow(xpt * xpt + ypt * ypt, 1.5);
        em = fmax(em, fabs(k1 - k0));
    }
    em = block_reduce_(em, 2, red);
    if (tid == 0) {
        *c.w.curv_err = em;
        *c.w.status = r.status;
        if (c.w.info) {
            mcq_info o;
            o.ipm_iters = r.ipm_iters;
            o.as_iters = r.as_iters;
            o.n_active_box = (int)nact;
            o.n_active_kappa = r.nact_kappa;
            o.kappa_max = r.km;
            o.kkt_res = c.fscale > 0.0 ? r.kkt / c.fscale : r.kkt;
            o.refine_rounds = c.refine_rounds;
            o.second_attempt = c.second_attempt;
            o.f32_factorisations = c.f32_count;
            o.gi_iters = r.gi_iters;
            c.tk[3] = TICK() - r.t_kernel0;
            c.tk[7] = TICK() - r.t_epi0;          // curvature check, (rare) curvature-row phase, outputs (ticks[7])
            c.tk[6] = (long long)clock64() - r.c_kernel0;
            for (int q = 0; q < 8; ++q) o.ticks[q] = c.tk[q];
            *(mcq_info*)c.w.info = o;
        }
    }
}

#include "mcq_gi.inc"

// One QP pass of problem pb, by its workgroup: the body of mcq_solve_kernel (and of every round of mcq_iqp_rounds_kernel).
__device__ __forceinline__ void solve_body(const McqBatch& B, const McqSet& IN, const int pb)
{
    const int tid = threadIdx.x;
    int n;
    double kbound, wveh;
    if (B.poison_lds) {      // debugging aid: whatever a phase reads from LDS without having written it shows up as NaN on every box
        for (int q = tid; q < SM_TOTAL; q += MCQ_NT) g_sm[q] = __longlong_as_double(-1LL);
        __syncthreads();
    }
    const long long t_kernel0 = TICK();
    const long long c_kernel0 = (long long)clock64();      // shader-clock counter (s_memtime): with ticks[3] the effective clock
    ctx_set_problem(B, IN, pb, n, kbound, wveh);
    const LCtx& c = G_CTX;
    ctx_set_solve(B, kbound);
    if (B.objective == MCQ_OBJ_SHORTEST_PATH) {
        if (*c.w.status != MCQ_OK) return;            // (written by mcq_assemble_sp_kernel)
    } else if (assemble_problem(c, wveh, nullptr, nullptr) != MCQ_OK) return;
    double* red = g_sm + SM_RED;
    const int nm = B.nmax;

    const gdouble* LO = VEC(c.w, nm, V_LO);
    const gdouble* HI = VEC(c.w, nm, V_HI);
    const gdouble* F = VEC(c.w, nm, V_F);
    gdouble* X = VEC(c.w, nm, V_X);
    gdouble* G = VEC(c.w, nm, V_G);
    gdouble* T0 = VEC(c.w, nm, V_T0);
    gdouble* T1 = VEC(c.w, nm, V_T1);
    gdouble* T2 = VEC(c.w, nm, V_T2);
    gdouble* T3 = VEC(c.w, nm, V_T3);
    gdouble* Q = VEC(c.w, nm, V_Q);
    gschar* ST = c.w.state;

    if (!c.direct) {   // f = F_SCALE E' k_ref
        gdouble* Fw = VEC(c.w, nm, V_F);
        tri_apply_Et(c, VEC(c.w, nm, V_KREF), Fw);        // (the pivots of T, V_IDL / V_TUC, are the assembly kernel's)
        for (int i = tid; i < n; i += MCQ_NT) Fw[i] *= MCQ_F_SCALE;
        __syncthreads();
    }
    problem_scales(c);
    // mcq_opts.algorithm = MCQ_ALG_GI: every problem through the Goldfarb-Idnani path at the end of this kernel, nothing else
    const bool gi_only = B.algorithm == MCQ_ALG_GI && !c.direct && (B.gi != nullptr || B.gis != nullptr);

    // ---- phase 1: box-constrained QP ---------------------------------------------------------------------------------------
    int ipm_iters = 0, as_iters = 0, nact_kappa = 0;
    double kkt = 0.0;
    const bool small = n <= IPB_E * MCQ_NT;
    // ---- warm start (IQP passes 2+): the working set of the previous pass, carried through the re-sampling by the glue kernel, is
    //      off by a few dozen rows; the one-row-per-neighbourhood exchange settles from it in a handful of rounds (each one
    //      factorisation + one solve) -- a third to two thirds of what interior point + exchange cost.  The vertex returned is an
    //      exact KKT point either way; if the exchange runs out of its rounds the cold path below takes over. ----
    bool warm_done = false;
    if (IN.warm && small && !gi_only) {
        const gschar* WS = (const gschar*)(IN.warm + (size_t)pb * nm);
        for (int i = tid; i < n; i += MCQ_NT)
            if (ST[i] == 0) { const signed char s = WS[i]; ST[i] = (s == 1 || s == -1) ? s : (signed char)0; }
        __syncthreads();
        const int capw = B.max_as_iter < MCQ_WARM_ROUNDS ? B.max_as_iter : MCQ_WARM_ROUNDS;
        const int sw = active_set(c, false, false, capw, 0, kappa_mem_lds(c));
        as_iters = c.out_iters;
        kkt = c.out_kkt;
        if (sw == MCQ_OK) warm_done = true;
        else {
            c.second_attempt = 2;       // reported: warm start abandoned
            for (int i = tid; i < n; i += MCQ_NT) {
                ST[i] = !(HI[i] - LO[i] > 1e-12) ? 2 : 0;
                X[i] = 0.5 * (LO[i] + HI[i]);
            }
            __syncthreads();
            gradient(c, X, nullptr, T0, G);
        }
    }
    const int as_warm = warm_done ? 0 : as_iters;
    int status = MCQ_OK;
    const long long t_ipm0 = TICK();
    if (gi_only) status = MCQ_ITER_CAP;
    else if (!warm_done) {
        status = small ? ipm_box(c, MCQ_IPM_TOL, false) : ipm(c, false);
        ipm_iters = c.out_iters;
        __syncthreads();      // (every thread has read what the phase left in the context before the next phase -- whose first act is to reset it --
                              //  starts: round 5, found on the SIMT interpreter, where a fibre that runs ahead made thread 0 report 0 iterations; on the
                              //  GPU a wave that far ahead of another was never observed, but `ipm_iters >= 1` below decides the Tapia rule per thread)
    }
    if (tid == 0) c.tk[4] = TICK() - t_ipm0;          // wall time of the interior-point phase (ticks[4])
    const long long t_as0 = TICK();
    if (warm_done) {
    } else if (status == MCQ_OK && small) {
        // Two attempts.  The pairs at mu = 1e-10 identify the active set of all but the degenerate / extremely
        // ill-conditioned instances (IQP passes on an already optimised raceline: dozens of bounds touched with multipliers
        // down to 1e-7 of the gradient scale, |x_ipm - x*| ~ 5 mm at that mu); there block pivoting from a guess that is off
        // by a few rows does not settle.  So the first active-set attempt is capped at a few rounds; if it runs out, the
        // interior point resumes from its own pairs down to mu = 1e-13 -- where the magnitude test separates those rows --
        // and the active-set phase starts again with the full budget.
        gdouble* XS = VEC(c.w, nm, V_TL);
        for (int i = tid; i < n; i += MCQ_NT) XS[i] = X[i];
        const int cap1 = B.max_as_iter < 6 ? B.max_as_iter : 6;
        status = active_set(c, false, ipm_iters >= 1 && c.last_step >= 0.9, cap1, 1, kappa_mem_lds(c));
        as_iters = c.out_iters;
        kkt = c.out_kkt;
        if (status == MCQ_ITER_CAP && B.max_as_iter > cap1) {
            c.second_attempt |= 1;
            for (int i = tid; i < n; i += MCQ_NT) X[i] = XS[i];
            __syncthreads();
            status = ipm_box(c, 1e-13, true);
            ipm_iters += c.out_iters;
            __syncthreads();
            if (status == MCQ_OK) {
                status = active_set(c, false, false, B.max_as_iter, 1, kappa_mem_lds(c));
                as_iters += c.out_iters;
                kkt = c.out_kkt;
            }
        }
    } else if (status == MCQ_OK) {
        status = active_set(c, false, ipm_iters >= 1 && c.last_step >= 0.9, B.max_as_iter, 1, kappa_mem_lds(c));
        as_iters = c.out_iters;
        kkt = c.out_kkt;
    }
    // ---- the shortest-path objective's own last resort (round 6; VERDICT r5 missing 7).  Its H is a cyclic tridiagonal with no factor E to change
    //      coordinates with, so the Goldfarb-Idnani path below cannot take it -- and does not have to: this QP has box rows only, and on the linear
    //      complementarity problem of a positive definite matrix with simple bounds the single-pivot backup rule of active_set() (Murty's
    //      largest-index rule, entered whenever full exchanges stop improving) terminates by theorem.  What could stop it short is the caller's
    //      round budget; a shortest-path problem that runs out of it goes on from its working set with a budget that only bounds the loop
    //      (a round is two scalar tridiagonal sweeps here, microseconds). ----
    if (c.direct && status == MCQ_ITER_CAP) {
        status = active_set(c, false, false, 8 * n + 200, 0, kappa_mem_lds(c));
        as_iters += c.out_iters;
        kkt = c.out_kkt;
        if (tid == 0) c.second_attempt |= 4;
    }
    as_iters += as_warm;        // rounds of an abandoned warm start are reported too
    if (tid == 0) c.tk[5] = TICK() - t_as0;           // wall time of the active-set phase (ticks[5])
    const long long t_epi0 = TICK();

    // kappa(alpha) = k_ref + E alpha
    for (int i = tid; i < n; i += MCQ_NT) X[i] = fmin(fmax(X[i], LO[i]), HI[i]);
    __syncthreads();
    double km = 0.0;
    bool dd_valid = false;
    if (!c.direct) {
        // (the second derivatives the post-check needs, D (n_x alpha) and D (n_y alpha), are by-products of this product: kept in two
        //  vectors only the curvature-row phase uses, so the post-check repeats the product only if that phase ran)
        tri_apply_E(c, X, VEC(c.w, nm, V_KREF), 1.0, T0, VEC(c.w, nm, V_TL), VEC(c.w, nm, V_TU));
        dd_valid = true;
        for (int i = tid; i < n; i += MCQ_NT) km = fmax(km, fabs(T0[i]));
        km = block_reduce_(km, 2, red);
    }

    // ---- phase 2 (rare): a curvature-bound row is violated at the box optimum -> interior point with the curvature rows,
    //      then the box active-set polish with the curvature multipliers frozen ---------------------------------------------
    if (status == MCQ_OK && B.check_kappa && !c.direct && km > kbound * (1.0 + 1e-9)) {
        dd_valid = false;
        status = ipm(c, true);
        ipm_iters += c.out_iters;
        __syncthreads();
        if (status == MCQ_OK) {
            status = active_set(c, true, false, B.max_as_iter, 1, kappa_mem_lds(c));
            as_iters += c.out_iters;
            kkt = c.out_kkt;
            nact_kappa = c.out_nk;
            if (status == MCQ_KAPPA_ACTIVE && B.kbig && B.kbig_slots > 0) {
                // More curvature rows in the working set than the LDS-resident arrays hold (MCQ_KMAX): quadprog has no such limit
                // [REF params/racecar.ini:49 curvlim].  The problem claims one of the handle's overflow slots -- MCQ_KBIG rows, the
                // Schur matrix and its elimination in HBM -- and the exchange goes on from the box working set the first
                // attempt left (its curvature flags are rebuilt from the interior point's pairs).  No
                // slot free (more than kbig_slots such problems at this moment): MCQ_KAPPA_NO_SLOT, taken up below.
                const int slot = claim_slot(B.slot_flags, B.kbig_slots, false);
                if (slot >= 0) {
                    status = active_set(c, true, false, B.max_as_iter, 0, kappa_mem_slot(B.kbig + (size_t)slot * MCQ_KBIG_SLOT));
                    as_iters += c.out_iters;
                    kkt = c.out_kkt;
                    nact_kappa = c.out_nk;
                    release_slot(B.slot_flags, slot);
                } else status = MCQ_KAPPA_NO_SLOT;        // not a verdict on the problem: the Goldfarb-Idnani path below takes it
            }
        }
        for (int i = tid; i < n; i += MCQ_NT) X[i] = fmin(fmax(X[i], LO[i]), HI[i]);
        __syncthreads();
        tri_apply_E(c, X, VEC(c.w, nm, V_KREF), 1.0, T0);
        double k2 = 0.0;
        for (int i = tid; i < n; i += MCQ_NT) k2 = fmax(k2, fabs(T0[i]));
        km = block_reduce_(k2, 2, red);
        if (status == MCQ_OK && km > kbound * (1.0 + 1e-8)) status = MCQ_KAPPA_ACTIVE;
    } else if (status == MCQ_OK && !B.check_kappa && !c.direct && km > kbound * (1.0 + 1e-9)) {
        status = MCQ_KAPPA_ACTIVE;      // curvature rows switched off by the caller: the violated row is reported, not enforced
    }

    // ---- the Goldfarb-Idnani path for whatever the phases above did not settle (mcq_gi.inc): quadprog's algorithm, finite by construction ----
    int gi_iters = 0;
    if (!c.direct && (B.gi || B.gis) && gi_eligible(status, B.check_kappa)) {
        status = gi_rescue(c, B, B.check_kappa != 0, status, as_iters, nact_kappa, kkt, km, gi_iters);
        dd_valid = true;
    }
    McqOutcome r;
    r.status = status; r.ipm_iters = ipm_iters; r.as_iters = as_iters; r.nact_kappa = nact_kappa; r.gi_iters = gi_iters;
    r.kkt = kkt; r.km = km; r.dd_valid = dd_valid;
    r.t_kernel0 = t_kernel0; r.c_kernel0 = c_kernel0; r.t_epi0 = t_epi0;
    write_outputs(c, r);
}

__global__ void __launch_bounds__(MCQ_NT, 2) mcq_solve_kernel(McqBatch B)
{
    solve_body(B, mcq_set_of(B), B.pb0 + (int)blockIdx.x);
}

// =====================================================================================================================
// K4: IQP glue -- re-linearisation on the device
// =====================================================================================================================
// Closed cubic spline with unit scalings through n points: the c-coefficients solve  circ(1, 4, 1) c = rhs,
// rhs_m = 3 (P_{m+1} - 2 P_m + P_{m-1}).  The inverse of circ(1,4,1) is known in closed form -- g_k = A lam^|k| with
// lam = sqrt(3) - 2, A = 1 / (4 + 2 lam), plus its periodic images -- so c is a short convolution: 0.268^48 ~ 4e-28, rings
// with more than 2 x 48 points use the truncated kernel, shorter rings the exact periodic one
// g_k = A (lam^k + lam^(n-k)) / (1 - lam^n).  RX / RY hold rhs; all threads of the block call.
#define RL_KW 48
__device__ void relin_spline_c(const gdouble* RX, const gdouble* RY, int n, gdouble* CX, gdouble* CY)
{
    const double lam = sqrt(3.0) - 2.0, A = 1.0 / (4.0 + 2.0 * lam);
    for (int i = threadIdx.x; i < n; i += MCQ_NT) {
        double cx, cy;
        if (n > 2 * RL_KW) {
            cx = A * RX[i];
            cy = A * RY[i];
            double g = A;
            int ju = i, jd = i;
            for (int k = 1; k <= RL_KW; ++k) {
                g *= lam;
                ju = ju + 1 == n ? 0 : ju + 1;
                jd = jd == 0 ? n - 1 : jd - 1;
                cx += g * (RX[ju] + RX[jd]);
                cy += g * (RY[ju] + RY[jd]);
            }
        } else {
            double ln = 1.0;
            for (int k = 0; k < n; ++k) ln *= lam;                       // lam^n
            const double sc = A / (1.0 - ln), il = 1.0 / lam;
            double pk = 1.0, qk = ln;                                    // lam^k, lam^(n-k)
            cx = 0.0;
            cy = 0.0;
            int j = i;
            for (int k = 0; k < n; ++k) {
                const double g = sc * (pk + qk);
                cx += g * RX[j];
                cy += g * RY[j];
                pk *= lam;
                qk *= il;
                j = j + 1 == n ? 0 : j + 1;
            }
        }
        CX[i] = cx;
        CY[i] = cy;
    }
}

// rhs of the c-system from the points
__device__ void relin_spline_rhs(const gdouble* PX, const gdouble* PY, int n, gdouble* RX, gdouble* RY)
{
    for (int i = threadIdx.x; i < n; i += MCQ_NT) {
        const int ip = i + 1 == n ? 0 : i + 1, im = i == 0 ? n - 1 : i - 1;
        RX[i] = 3.0 * ((PX[ip] - PX[i]) - (PX[i] - PX[im]));
        RY[i] = 3.0 * ((PY[ip] - PY[i]) - (PY[i] - PY[im]));
    }
}

// Front half of the glue, shared by mcq_relinearise_kernel and mcq_raceline_kernel: raceline p + a n (and the shifted
// widths), closed unit-scaling spline through it, spline lengths and their running sum, point count of the re-sampled ring.
// vec slots: 0/1 raceline points (a-coefficients), 2/3 rhs, 4/5 c-coefficients, 6 lengths, 7 running sum, 8/9 widths.
// Returns the number of points kept (tph.interp_splines, incl_last_point = False) or -1; `total` = length of the raceline.
// lds: RELIN_LDS doubles of the caller's LDS (the glue kernels bring their own 16 KB; mcq_iqp_rounds_kernel lends the solver's array).
#define RELIN_LDS 2050
__device__ __forceinline__ int relin_front(const gdouble* ref, const gdouble* nv, const gdouble* al, int n, double alpha_scale,
                                           double stepsize, gdouble* vec, size_t nm, double& total, double* lds)
{
    double* sbuf = lds;
    double& s_carry = lds[2048];
    int& s_m = *(int*)(lds + 2049);
    const int tid = threadIdx.x;
    gdouble* PX = vec + 0 * nm;   gdouble* PY = vec + 1 * nm;    // raceline points (a-coefficients)
    gdouble* RX = vec + 2 * nm;   gdouble* RY = vec + 3 * nm;    // rhs, later the new points
    gdouble* CX = vec + 4 * nm;   gdouble* CY = vec + 5 * nm;    // c-coefficients
    gdouble* LEN = vec + 6 * nm;  gdouble* CUM = vec + 7 * nm;   // spline lengths, their running sum
    gdouble* WR = vec + 8 * nm;   gdouble* WL = vec + 9 * nm;    // shifted track widths

    // ---- raceline and shifted widths:  p + a n,  w_right - a,  w_left + a   (a = damped alpha) --------------------------
    for (int i = tid; i < n; i += MCQ_NT) {
        const double a = alpha_scale * al[i];
        PX[i] = ref[4 * i] + a * nv[2 * i];
        PY[i] = ref[4 * i + 1] + a * nv[2 * i + 1];
        WR[i] = ref[4 * i + 2] - a;
        WL[i] = ref[4 * i + 3] + a;
    }
    __syncthreads();
    relin_spline_rhs(PX, PY, n, RX, RY);
    __syncthreads();
    relin_spline_c(RX, RY, n, CX, CY);
    __syncthreads();

    // ---- spline lengths: 15 points per segment, sum of the 14 chords (tph.calc_spline_lengths) --------------------------
    for (int i = tid; i < n; i += MCQ_NT) {
        const int ip = i + 1 == n ? 0 : i + 1;
        const double ax = PX[i], ay = PY[i], cx = CX[i], cy = CY[i];
        const double dlx = PX[ip] - ax, dly = PY[ip] - ay;
        const double bx = dlx - (2.0 * cx + CX[ip]) / 3.0, by = dly - (2.0 * cy + CY[ip]) / 3.0;
        const double dx = (CX[ip] - cx) / 3.0, dy = (CY[ip] - cy) / 3.0;
        double len = 0.0, x0 = ax, y0 = ay;
        for (int k = 1; k < 15; ++k) {
            const double t = (double)k / 14.0;
            const double x1 = ax + bx * t + cx * t * t + dx * t * t * t;
            const double y1 = ay + by * t + cy * t * t + dy * t * t * t;
            len += hypot(x1 - x0, y1 - y0);
            x0 = x1;
            y0 = y1;
        }
        LEN[i] = len;
    }
    __syncthreads();
    // ---- running sum, in the summation order of numpy.cumsum (the point count below is a ceil() of its last entry):
    //      2048-entry chunks through LDS, one thread adds sequentially ---------------------------------------------------
    if (tid == 0) s_carry = 0.0;
    for (int c0 = 0; c0 < n; c0 += 2048) {
        const int cn = n - c0 < 2048 ? n - c0 : 2048;
        __syncthreads();
        for (int q = tid; q < cn; q += MCQ_NT) sbuf[q] = LEN[c0 + q];
        __syncthreads();
        if (tid == 0) {
            double acc = s_carry;
            for (int q = 0; q < cn; ++q) { acc += sbuf[q]; sbuf[q] = acc; }
            s_carry = acc;
        }
        __syncthreads();
        for (int q = tid; q < cn; q += MCQ_NT) CUM[c0 + q] = sbuf[q];
    }
    __syncthreads();
    total = s_carry;
    if (tid == 0) {
        const double cnt = ceil(total / stepsize) + 1.0;         // no_interp_points of tph.interp_splines
        s_m = (cnt >= 2.0 && cnt <= 2.0e9) ? (int)cnt - 1 : -1;  // points kept (incl_last_point = False)
    }
    __syncthreads();
    return s_m;
}

// The glue of one track, by its workgroup: the body of mcq_relinearise_kernel (and of every round of mcq_iqp_rounds_kernel).  IO: the ring
// buffers of this round (the launch's own -- R -- for the kernel); lds: RELIN_LDS doubles.
struct McqRelinIO {
    const int* n_in; const double* ref_in; const double* nv_in;
    double* ref_out; double* nv_out; int* n_out;
    double alpha_scale;
};
__device__ __forceinline__ void relin_body(const McqRelin& R, const McqRelinIO& IO, const int pb, double* lds)
{
    const int tid = threadIdx.x;
    if (R.live && R.live[pb] == 0) return;
    const size_t nm = (size_t)R.nmax;
    const int n = IO.n_in[pb];
    const gdouble* ref = (const gdouble*)(IO.ref_in + (size_t)pb * nm * 4);
    const gdouble* nv = (const gdouble*)(IO.nv_in + (size_t)pb * nm * 2);
    const gdouble* al = (const gdouble*)(R.alpha + (size_t)pb * nm);
    gdouble* refo = (gdouble*)(IO.ref_out + (size_t)pb * nm * 4);
    gdouble* nvo = (gdouble*)(IO.nv_out + (size_t)pb * nm * 2);
    gdouble* vec = (gdouble*)(R.vec + (size_t)pb * nm * MCQ_NVEC);
    gdouble* PX = vec + 0 * nm;   gdouble* PY = vec + 1 * nm;    // raceline points (a-coefficients)
    gdouble* CX = vec + 4 * nm;   gdouble* CY = vec + 5 * nm;    // c-coefficients
    gdouble* LEN = vec + 6 * nm;  gdouble* CUM = vec + 7 * nm;   // spline lengths, their running sum
    gdouble* WR = vec + 8 * nm;   gdouble* WL = vec + 9 * nm;    // shifted track widths
    gdouble* QX = vec + 10 * nm;  gdouble* QY = vec + 11 * nm;   // re-sampled points
    gint* n_out = (gint*)(IO.n_out + pb);
    gint* status = (gint*)(R.status + pb);
    if (n < 3) {
        if (tid == 0) { *status = MCQ_BAD_INPUT; *n_out = n; }
        return;
    }
    double total;
    const int m = relin_front(ref, nv, al, n, IO.alpha_scale, R.stepsize, vec, nm, total, lds);
    if (m < 3 || m > R.nmax) {
        if (tid == 0) { *status = MCQ_BAD_INPUT; *n_out = n; }
        return;
    }

    // ---- re-sampling: numpy.linspace(0, total, m + 1)[:m], searchsorted(cum, q, side='right'), Horner evaluation;
    //      widths linearly between the segment's end points (closed) ---------------------------------------------------------
    const double step = total / (double)m;
    for (int j = tid; j < m; j += MCQ_NT) {
        const double q = (double)j * step;
        int lo = 0, hi = n;                                      // first index with CUM[idx] > q
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (CUM[mid] > q) hi = mid; else lo = mid + 1;
        }
        const int s = lo < n - 1 ? lo : n - 1;
        const int sp = s + 1 == n ? 0 : s + 1;
        const double start = s > 0 ? CUM[s - 1] : 0.0;
        const double t = (q - start) / LEN[s];
        const double ax = PX[s], ay = PY[s], cx = CX[s], cy = CY[s];
        const double bx = (PX[sp] - ax) - (2.0 * cx + CX[sp]) / 3.0, by = (PY[sp] - ay) - (2.0 * cy + CY[sp]) / 3.0;
        const double dx = (CX[sp] - cx) / 3.0, dy = (CY[sp] - cy) / 3.0;
        QX[j] = ax + t * (bx + t * (cx + t * dx));
        QY[j] = ay + t * (by + t * (cy + t * dy));
        refo[4 * j + 2] = WR[s] + (WR[sp] - WR[s]) * t;
        refo[4 * j + 3] = WL[s] + (WL[sp] - WL[s]) * t;
        if (R.state_out) {      // working set of the pass just solved, carried to the new ring: the nearer end of the segment
            const signed char so = ((const gschar*)(R.state_in + (size_t)pb * nm))[t > 0.5 ? sp : s];
            ((gschar*)(R.state_out + (size_t)pb * nm))[j] = (so == 1 || so == -1) ? so : (signed char)0;
        }
    }
    __syncthreads();
    for (int j = tid; j < m; j += MCQ_NT) {
        refo[4 * j] = QX[j];
        refo[4 * j + 1] = QY[j];
    }
    // ---- normals of the closed unit-scaling spline through the new ring:  (b_y, -b_x) / |b| ------------------------------
    relin_spline_rhs(QX, QY, m, PX, PY);                         // PX / PY reused as rhs
    __syncthreads();
    relin_spline_c(PX, PY, m, CX, CY);
    __syncthreads();
    for (int j = tid; j < m; j += MCQ_NT) {
        const int jp = j + 1 == m ? 0 : j + 1;
        const double bx = (QX[jp] - QX[j]) - (2.0 * CX[j] + CX[jp]) / 3.0;
        const double by = (QY[jp] - QY[j]) - (2.0 * CY[j] + CY[jp]) / 3.0;
        const double nrm = sqrt(by * by + bx * bx);
        nvo[2 * j] = by / nrm;
        nvo[2 * j + 1] = -bx / nrm;
    }
    if (tid == 0) { *status = MCQ_OK; *n_out = m; }
}

__global__ void __launch_bounds__(MCQ_NT) mcq_relinearise_kernel(McqRelin R)
{
    __shared__ double lds[RELIN_LDS];
    McqRelinIO io;
    io.n_in = R.n_in; io.ref_in = R.ref_in; io.nv_in = R.nv_in; io.ref_out = R.ref_out; io.nv_out = R.nv_out; io.n_out = R.n_out;
    io.alpha_scale = R.alpha_scale;
    relin_body(R, io, (int)blockIdx.x, lds);
}

// ---- iqp_handler's bookkeeping between the passes (see McqIqpStep), one track: phase, round, ring buffer in use and the two size arrays
//      are the caller's (the kernel's launch parameters; the round loop of mcq_iqp_rounds_kernel); count: keep S.live_count ------------
__device__ __forceinline__ void iqp_step_track(const McqIqpStep& S, int k, int phase, int round, int cur, const int* n_ring, int* n_next, bool count)
{
    if (phase == 0) {
        if (S.live[k] == 0) return;
        const int st = S.status[k];
        const double ce = S.curv[k];
        S.final_n[k] = n_ring[k];
        S.final_buf[k] = cur;
        S.final_curv[k] = ce;
        S.final_status[k] = st;
        S.final_rounds[k] = round;
        if (S.curv_trace && round <= MCQ_IQP_TRACE) S.curv_trace[(size_t)k * MCQ_IQP_TRACE + round - 1] = ce;
        const bool stop = st != MCQ_OK || (round >= S.iters_min && ce <= S.curv_allowed);
        if (stop) S.live[k] = 0;
        else if (count) atomicAdd(S.live_count, 1);
    } else {
        if (S.live[k] != 0 && S.relin_status[k] != MCQ_OK) {      // the re-sampled ring does not fit the buffers
            S.live[k] = 0;
            S.final_status[k] = MCQ_RING_OVERFLOW;
            if (count) atomicAdd(S.live_count, -1);
        }
        if (S.live[k] == 0) n_next[k] = 0;
    }
}

__global__ void __launch_bounds__(256) mcq_iqp_step_kernel(McqIqpStep S)
{
    const int k = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (k >= S.batch) return;
    iqp_step_track(S, k, S.phase, S.round, S.cur, S.n_ring, S.n_next, true);
}

// ---- The first rounds of iqp_handler for a batch of tracks as ONE launch: a track's rounds -- QP pass, termination test, glue -- depend on
//      nothing but its own previous round, and no track can end before round iters_min, so a workgroup takes its track through `rounds`
//      rounds on its own (the bodies of mcq_solve_kernel, mcq_iqp_step_kernel and mcq_relinearise_kernel, between workgroup barriers) and
//      nobody waits for anybody: launched round by round, every round ended with its slowest track (a warm start that falls back takes
//      8.7 ms where the median track takes 3) while most compute units idled.  F.live_count: tracks still iterating after the last round. ----
__global__ void __launch_bounds__(MCQ_NT, 2) mcq_iqp_rounds_kernel(McqIqpRounds F)
{
    const int pb = (int)blockIdx.x;
    for (int it = 1; it <= F.rounds; ++it) {
        const int cur = (it - 1) & 1;
        if (F.S.live[pb] == 0) {
            // (the track stopped: neither ring buffer holds a ring to solve any more -- the round-by-round loop that may follow skips it)
            if (threadIdx.x == 0) { F.n_set[0][pb] = 0; F.n_set[1][pb] = 0; }
            break;
        }
        McqSet in;
        in.n_list = F.n_set[cur];
        in.ref = F.ref_set[cur];
        in.nv = F.nv_set[cur];
        in.sc = it == 1 ? F.sc : nullptr;                       // the re-spline of passes 2+ uses unit scalings (upstream)
        in.warm = it > 1 ? F.warm : nullptr;
        solve_body(F.B, in, pb);
        __syncthreads();
        if (threadIdx.x == 0) iqp_step_track(F.S, pb, 0, it, cur, F.n_set[cur], F.n_set[1 - cur], it == F.rounds);
        __syncthreads();
        McqRelinIO io;
        io.n_in = F.n_set[cur]; io.ref_in = F.ref_set[cur]; io.nv_in = F.nv_set[cur];
        io.ref_out = F.ref_set[1 - cur]; io.nv_out = F.nv_set[1 - cur]; io.n_out = F.n_set[1 - cur];
        io.alpha_scale = it < F.S.iters_min ? (double)it / (double)F.S.iters_min : 1.0;
        relin_body(F.R, io, pb, g_sm);
        __syncthreads();
        if (threadIdx.x == 0) iqp_step_track(F.S, pb, 1, it, cur, F.n_set[cur], F.n_set[1 - cur], it == F.rounds);
        __syncthreads();
    }
}

// =====================================================================================================================
// K5: ggv velocity profile and lap time of many variants
// =====================================================================================================================
// numpy.interp: piecewise linear, clamped at both ends; xs ascending, stride 3 (ggv rows) or 2 (machine rows)
__device__ __forceinline__ double vp_interp(double x, const gdouble* tab, int cnt, int stride, int col)
{
    if (x <= tab[0]) return tab[col];
    if (x >= tab[(size_t)(cnt - 1) * stride]) return tab[(size_t)(cnt - 1) * stride + col];
    int k = 1;
    while (k < cnt - 1 && tab[(size_t)k * stride] <= x) ++k;                 // tab[k-1].x <= x < tab[k].x
    const double x0 = tab[(size_t)(k - 1) * stride], x1 = tab[(size_t)k * stride];
    const double y0 = tab[(size_t)(k - 1) * stride + col], y1 = tab[(size_t)k * stride + col];
    return y0 + (y1 - y0) * (x - x0) / (x1 - x0);
}

// longitudinal acceleration still available at speed v on radius `rad` with friction coefficient mu (tph.calc_vel_profile.calc_ax_poss:
// tyre potential shared with the lateral acceleration through the friction ellipse of exponent e, machine limit when accelerating,
// drag -- which helps when the deceleration is integrated backwards)
__device__ __forceinline__ double vp_ax_possible(double v, double rad, double mu, const gdouble* ggv, int ng, const gdouble* axm, int nam,
                                                 bool accel, double e, double drag_over_m)
{
    const double ax_t = fabs(mu * vp_interp(v, ggv, ng, 3, 1)), ay_t = mu * vp_interp(v, ggv, ng, 3, 2);
    const double ay_used = v * v / rad;
    const double radicand = 1.0 - pow(ay_used / ay_t, e);
    const double ax_tires = radicand > 0.0 ? ax_t * pow(radicand, 1.0 / e) : 0.0;
    const double ax_drag = -v * v * drag_over_m;
    if (accel) return fmin(ax_tires, vp_interp(v, axm, nam, 2, 1)) + ax_drag;
    return ax_tires - ax_drag;
}

// One thread per (track, vehicle) variant; the lap-doubled profile lives in a variant-minor scratch (S[j * batch]: coalesced
// across the threads of a wave).  Follows tph.calc_vel_profile step by step (restated in oracle/vel_ref.py, the checker):
//   lateral limit: fixed point of v = sqrt(ay_max(v) R) over ALL ggv rows, at most 100 rounds, stopped when the largest relative
//     change is below 0.5 % (a NaN in that maximum -- kappa == 0 gives inf / inf -- never passes, as numpy.max behaves);
//   forward sweep over two laps, backward sweep over the doubled SECOND lap of the forward result, second lap of that out;
//   a sweep is switched on at the first point of every acceleration phase of the profile it starts from (v[i+1] > v[i] and not
//     v[i] > v[i-1], both on the values before the sweep touched them) and switched off where the attainable speed exceeds v_max;
//   the backward sweep re-evaluates the deceleration one point ahead at the attainable speed and keeps the smaller speed;
//   backward, step p -> p-1 uses el[p] (upstream flips its arrays as a whole);
//   optional: a friction coefficient per waypoint (V.mu: scales both tyre limits; the first estimate of the lateral limit uses its MEAN, as
//     upstream -- the fixed point stops on a 0.5 % change, so the start matters at that level) and tph.conv_filt's closed moving average
//     of odd width V.filt_window over the finished profile [REF main_globaltraj.py:407, params/racecar.ini:54-57], lap time from the
//     filtered profile.
// A ggv / machine table that ends below v_max (upstream raises RuntimeError) gives lap_time = NaN; so does a filter window that is even
// or wider than the ring.
__global__ void __launch_bounds__(64) mcq_vel_profile_kernel(McqVel V)
{
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= V.batch) return;
    const int bt = V.batch;
    const int trk = V.track_of ? V.track_of[v] : v;
    const int n = V.n_of_track ? V.n_of_track[trk] : V.n;        // ragged tracks: per-row waypoint counts
    // a variant that cannot be computed (bad row length, or -- below -- a ggv / machine table that ends under its v_max, where tph
    // raises) is flagged by lap_time = NaN AND a vx_out row of NaNs: a C-ABI caller never sees stale buffer contents as a profile
    if (n < 2 || n > V.nmax) {
        V.lap_time[v] = NAN;
        for (int i = 0; i < V.nmax; ++i) V.vx_out[(size_t)v * V.nmax + i] = NAN;
        return;
    }
    const size_t row = (size_t)trk * V.nmax;
    const gdouble* kap = (const gdouble*)(V.kappa + row);
    const gdouble* el = (const gdouble*)(V.el + row);
    const gdouble* ggv = (const gdouble*)(V.ggv + (size_t)v * V.ng * 3);
    const gdouble* axm = (const gdouble*)(V.axm + (size_t)v * V.nam * 2);
    const gdouble* muv = V.mu ? (const gdouble*)(V.mu + row) : nullptr;
    const int fw = V.filt_window > 1 ? V.filt_window : 0;
    gdouble* S = (gdouble*)V.scratch + v;       // S[j * bt]
    const double vmax = V.vmax[v], e = V.dyn_exp, dom = V.drag[v] / V.mass[v];
    const int ng = V.ng, nam = V.nam;
    if (ggv[(size_t)(ng - 1) * 3] < vmax || axm[(size_t)(nam - 1) * 2] < vmax || (fw && (fw % 2 == 0 || fw > n))) {
        V.lap_time[v] = NAN;
        for (int i = 0; i < n; ++i) V.vx_out[(size_t)v * V.nmax + i] = NAN;
        return;
    }
    double aymin = ggv[2];
    for (int k = 1; k < ng; ++k) aymin = fmin(aymin, ggv[(size_t)k * 3 + 2]);
#define VP_RAD(i_) ((kap[(i_)] != 0.0) ? fabs(1.0 / kap[(i_)]) : (double)INFINITY)
#define VP_MU(i_) (muv ? muv[(i_)] : 1.0)

    // ---- lateral limit ------------------------------------------------------------------------------------------------------------
    double mu_mean = 1.0;
    if (muv) {
        mu_mean = 0.0;
        for (int i = 0; i < n; ++i) mu_mean += muv[i];
        mu_mean /= (double)n;
    }
    for (int i = 0; i < n; ++i) S[(size_t)i * bt] = sqrt(mu_mean * aymin * VP_RAD(i));
    for (int it = 0; it < 100; ++it) {
        double worst = 0.0;
        bool any_nan = false;
        for (int i = 0; i < n; ++i) {
            const double vx = S[(size_t)i * bt];
            const double vn = sqrt(VP_MU(i) * vp_interp(vx, ggv, ng, 3, 2) * VP_RAD(i));
            const double ch = fabs(vn / vx - 1.0);
            if (ch != ch) any_nan = true;
            worst = fmax(worst, ch);
            S[(size_t)i * bt] = vn;
        }
        if (!any_nan && worst < 0.005) break;
    }
    // ---- lap doubled, cut at the top speed ------------------------------------------------------------------------------------------
    for (int i = 0; i < n; ++i) {
        const double vx = fmin(S[(size_t)i * bt], vmax);
        S[(size_t)i * bt] = vx;
        S[(size_t)(n + i) * bt] = vx;
    }
    const int m2 = 2 * n;
    // ---- forward: acceleration-limited --------------------------------------------------------------------------------------------
    {
        bool active = false, prev_rising = false;
        double cur = S[0];                       // v[j] as the sweep left it
        double o_cur = cur;                      // v[j] before the sweep
        for (int j = 0; j < m2 - 1; ++j) {
            const int i = j < n ? j : j - n;
            double nxt = S[(size_t)(j + 1) * bt];   // untouched so far: the value before the sweep
            const bool rising = nxt > o_cur;
            if (rising && !prev_rising) active = true;
            prev_rising = rising;
            o_cur = nxt;
            if (active) {
                const double ax = vp_ax_possible(cur, VP_RAD(i), VP_MU(i), ggv, ng, axm, nam, true, e, dom);
                const double vnext = sqrt(cur * cur + 2.0 * ax * el[i]);
                if (vnext < nxt) { nxt = vnext; S[(size_t)(j + 1) * bt] = nxt; }
                if (vnext > vmax) active = false;
            }
            cur = nxt;
        }
    }
    // second lap of the forward result, doubled
    for (int i = 0; i < n; ++i) S[(size_t)i * bt] = S[(size_t)(n + i) * bt];
    // ---- backward: deceleration-limited, in flipped order  q = m2 - 1 - j ------------------------------------------------------------
    {
        bool active = false, prev_rising = false;
        double cur = S[(size_t)(m2 - 1) * bt];
        double o_cur = cur;
        for (int j = m2 - 1; j >= 1; --j) {
            const int i = j < n ? j : j - n;                 // point left
            const int im = (j - 1) < n ? j - 1 : j - 1 - n;  // point reached
            double prv = S[(size_t)(j - 1) * bt];
            const bool rising = prv > o_cur;
            if (rising && !prev_rising) active = true;
            prev_rising = rising;
            o_cur = prv;
            if (active) {
                const double c2 = cur * cur, l2 = 2.0 * el[i];
                const double ax = vp_ax_possible(cur, VP_RAD(i), VP_MU(i), ggv, ng, axm, nam, false, e, dom);
                double vprev = sqrt(c2 + ax * l2);
                const double ax2 = vp_ax_possible(vprev, VP_RAD(im), VP_MU(im), ggv, ng, axm, nam, false, e, dom);
                const double vtmp = sqrt(c2 + ax2 * l2);
                if (vtmp < vprev) vprev = vtmp;
                if (vprev < prv) { prv = vprev; S[(size_t)(j - 1) * bt] = prv; }
                if (vprev > vmax) active = false;
            }
            cur = prv;
        }
    }
#undef VP_RAD
#undef VP_MU
    // ---- tph.conv_filt, closed: the mean over fw consecutive points of the ring, products summed in numpy.convolve's order; from the
    //      second lap into the first (done with), which then is the profile ------------------------------------------------------------
    size_t lap = (size_t)n;
    if (fw) {
        const int hw = (fw - 1) / 2;
        const double ker = 1.0 / (double)fw;
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int d = hw; d >= -hw; --d) {
                int j = i + d;
                j = j < 0 ? j + n : j >= n ? j - n : j;
                acc += S[(size_t)(n + j) * bt] * ker;
            }
            S[(size_t)i * bt] = acc;
        }
        lap = 0;
    }
    // ---- the profile out; lap time from the piecewise-constant accelerations (tph.calc_ax_profile / calc_t_profile) -----------
    gdouble* out = (gdouble*)(V.vx_out + (size_t)v * V.nmax);
    double t = 0.0;
    const double v0 = S[lap * bt];
    double va = v0;
    for (int i = 0; i < n; ++i) {
        const double vb = i + 1 < n ? S[(lap + i + 1) * bt] : v0;
        out[i] = va;
        // constant acceleration over the element: t = 2 l / (v_a + v_b).  Algebraically tph.calc_t_profile's
        // (-v_a + sqrt(v_a^2 + 2 a l)) / a with a = (v_b^2 - v_a^2) / (2 l), which cancels catastrophically as a -> 0 (a 1e-15
        // ripple on a speed-limited stretch moves that expression by 0.1 s per element); this form has no such case.
        t += 2.0 * el[i] / (va + vb);
        va = vb;
    }
    V.lap_time[v] = t;
}

// ---- what main_globaltraj.py does with alpha before the velocity profile [REF main_globaltraj.py:371-387]: tph.create_raceline
//      (raceline p + alpha n, closed unit-scaling spline, re-sampling at ~stepsize_interp_after_opt) followed by
//      tph.calc_head_curv_an (heading and curvature from the spline's derivatives, analytically) -- one workgroup per track,
//      outputs strided by mmax: exactly what mcq_vel_profile_kernel consumes (kappa, el_lengths). ----
__global__ void __launch_bounds__(MCQ_NT) mcq_raceline_kernel(McqRace Q)
{
    const int tid = threadIdx.x, pb = blockIdx.x;
    const size_t nm = (size_t)Q.nmax, mm = (size_t)Q.mmax;
    const int n = Q.n_in ? Q.n_in[pb] : Q.nmax;
    const gdouble* ref = (const gdouble*)(Q.ref + (size_t)pb * nm * 4);
    const gdouble* nv = (const gdouble*)(Q.nv + (size_t)pb * nm * 2);
    const gdouble* al = (const gdouble*)(Q.alpha + (size_t)pb * nm);
    gdouble* vec = (gdouble*)(Q.vec + (size_t)pb * nm * MCQ_NVEC);
    gdouble* PX = vec + 0 * nm;   gdouble* PY = vec + 1 * nm;
    gdouble* CX = vec + 4 * nm;   gdouble* CY = vec + 5 * nm;
    gdouble* LEN = vec + 6 * nm;  gdouble* CUM = vec + 7 * nm;
    gint* m_out = (gint*)(Q.m_out + pb);
    gint* status = (gint*)(Q.status + pb);
    if (n < 3) {
        if (tid == 0) { *status = MCQ_BAD_INPUT; *m_out = 0; }
        return;
    }
    double total;
    __shared__ double relin_lds[RELIN_LDS];
    const int m = relin_front(ref, nv, al, n, 1.0, Q.stepsize, vec, nm, total, relin_lds);
    if (m < 2 || m > Q.mmax) {
        if (tid == 0) { *status = MCQ_BAD_INPUT; *m_out = m > 0 ? m : 0; }
        return;
    }
    gdouble* xy = Q.xy_out ? (gdouble*)(Q.xy_out + (size_t)pb * mm * 2) : nullptr;
    gdouble* psi = Q.psi_out ? (gdouble*)(Q.psi_out + (size_t)pb * mm) : nullptr;
    gdouble* kap = (gdouble*)(Q.kappa_out + (size_t)pb * mm);
    gdouble* el = (gdouble*)(Q.el_out + (size_t)pb * mm);
    const double step = total / (double)m;                       // numpy.linspace(0, total, m + 1): station j = j * step
    const double pi = 3.14159265358979323846;
    for (int j = tid; j < m; j += MCQ_NT) {
        const double q = (double)j * step;
        int lo = 0, hi = n;                                      // first index with CUM[idx] > q
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (CUM[mid] > q) hi = mid; else lo = mid + 1;
        }
        const int s = lo < n - 1 ? lo : n - 1;
        const int sp = s + 1 == n ? 0 : s + 1;
        const double start = s > 0 ? CUM[s - 1] : 0.0;
        const double t = (q - start) / LEN[s];
        const double ax = PX[s], ay = PY[s], cx = CX[s], cy = CY[s];
        const double bx = (PX[sp] - ax) - (2.0 * cx + CX[sp]) / 3.0, by = (PY[sp] - ay) - (2.0 * cy + CY[sp]) / 3.0;
        const double dx = (CX[sp] - cx) / 3.0, dy = (CY[sp] - cy) / 3.0;
        if (xy) {
            xy[2 * j] = ax + t * (bx + t * (cx + t * dx));
            xy[2 * j + 1] = ay + t * (by + t * (cy + t * dy));
        }
        const double xd = bx + 2.0 * cx * t + 3.0 * dx * t * t, yd = by + 2.0 * cy * t + 3.0 * dy * t * t;
        const double xdd = 2.0 * cx + 6.0 * dx * t, ydd = 2.0 * cy + 6.0 * dy * t;
        if (psi) {                                               // heading, 0 = north, wrapped to [-pi, pi) (tph.normalize_psi)
            double a = atan2(yd, xd) - 0.5 * pi;
            if (a >= pi) a -= 2.0 * pi;
            else if (a < -pi) a += 2.0 * pi;
            psi[j] = a;
        }
        const double v2 = xd * xd + yd * yd;
        kap[j] = (xd * ydd - yd * xdd) / (v2 * sqrt(v2));
        el[j] = j + 1 < m ? (double)(j + 1) * step - q : total - q;
    }
    if (tid == 0) { *status = MCQ_OK; *m_out = m; }
}

// ---- tph.check_normals_crossing as prep_track calls it [REF helper_funcs_glob/src/prep_track.py:57-59]: do the normal segments
//      [p - w_left n, p + w_right n] of two waypoints at most `horizon` apart intersect?  One workgroup per track, thread per
//      waypoint, 2x2 system per pair in closed form (parallel normals: no crossing). ----
__global__ void __launch_bounds__(MCQ_NT) mcq_normals_crossing_kernel(int nmax, const int* n_list, const double* ref_all,
                                                                      const double* nv_all, int horizon, int* crossing_out)
{
    __shared__ int s_hit;
    const int tid = threadIdx.x, pb = blockIdx.x;
    const int n = n_list ? n_list[pb] : nmax;
    const gdouble* ref = (const gdouble*)(ref_all + (size_t)pb * nmax * 4);
    const gdouble* nv = (const gdouble*)(nv_all + (size_t)pb * nmax * 2);
    if (tid == 0) s_hit = 0;
    __syncthreads();
    if (n < 2 || horizon >= n) {                                 // tph raises: "Horizon ... is too large for a track with ..."
        if (tid == 0) ((gint*)crossing_out)[pb] = -1;
        return;
    }
    int hit = 0;
    for (int i = tid; i < n; i += MCQ_NT) {
        const double px = ref[4 * i], py = ref[4 * i + 1], wr = ref[4 * i + 2], wl = ref[4 * i + 3];
        const double ax = nv[2 * i], ay = nv[2 * i + 1];
        for (int d = 1; d <= horizon; ++d) {
            const int j = i + d < n ? i + d : i + d - n;
            const double bx = nv[2 * j], by = nv[2 * j + 1];
            // upstream skips neighbours whose normal is collinear: numpy.isclose(cross(n_i, n_j), 0.0), i.e. |cross| <= 1e-8
            if (fabs(ax * by - ay * bx) <= 1e-8) continue;
            const double rx = ref[4 * j] - px, ry = ref[4 * j + 1] - py;
            const double det = -ax * by + ay * bx;               // p_i + l0 n_i = p_j + l1 n_j
            const double l0 = (-rx * by + ry * bx) / det, l1 = (ax * ry - ay * rx) / det;
            // both parameters inside [-w_left, w_right] of their point, bounds included (as upstream)
            if (l0 >= -wl && l0 <= wr && l1 >= -ref[4 * j + 3] && l1 <= ref[4 * j + 2]) hit = 1;
        }
    }
    if (hit) s_hit = 1;
    __syncthreads();
    if (tid == 0) ((gint*)crossing_out)[pb] = s_hit;
}

// ---- fp32 boundary (BASELINE config 5): tracks and results stored as float in HBM, every bit of arithmetic still fp64.
//      cond(H) = 1e9...1e12 rules out fp32 factors; what fp32 buys is half the bytes at the
//      boundary -- the rows coming in and, above all, the alpha vectors that go through the all-gather.  Plain streaming
//      kernels: 16-byte loads on the narrow side, grid-stride. ----
__global__ void mcq_widen_kernel(const float* src, double* dst, size_t count)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t quads = count / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride) {
        const float a = src[4 * q], b = src[4 * q + 1], c = src[4 * q + 2], d = src[4 * q + 3];
        dst[4 * q] = (double)a;
        dst[4 * q + 1] = (double)b;
        dst[4 * q + 2] = (double)c;
        dst[4 * q + 3] = (double)d;
    }
    for (size_t i = 4 * quads + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = (double)src[i];
}

__global__ void mcq_narrow_kernel(const double* src, float* dst, size_t count)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t quads = count / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += stride) {
        const double a = src[4 * q], b = src[4 * q + 1], c = src[4 * q + 2], d = src[4 * q + 3];
        dst[4 * q] = (float)a;
        dst[4 * q + 1] = (float)b;
        dst[4 * q + 2] = (float)c;
        dst[4 * q + 3] = (float)d;
    }
    for (size_t i = 4 * quads + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = (float)src[i];
}

// ---- fp32 rows, increment layout (include/mcq.h: MCQ_F32_INCREMENTS): the QP sees the reference line only through differences of
//      neighbouring waypoints, so the float rows carry those -- row i = [x_{i+1} - x_i, y_{i+1} - y_i, w_r, w_l], ring order -- and
//      x, y are rebuilt here in fp64: running sum from the track's origin, with the closure defect of the float increments (they do
//      not sum to zero exactly) spread evenly over the ring.  One wave per track: lane l sums its contiguous slice, an exclusive
//      scan over the 64 slice sums (wave shuffles), then every lane writes its slice.  layout 0 (absolute rows): widened, origin
//      added.  32 KB per track next to the ~180 MB the solver moves for it: not measurable. ----
__global__ void __launch_bounds__(64) mcq_widen_rows_kernel(const float* rows, const double* origin, double* dst, int n, int layout)
{
    const int pb = blockIdx.x, lane = threadIdx.x & 63;
    const float* src = rows + (size_t)pb * n * 4;
    double* out = dst + (size_t)pb * n * 4;
    const double ox = origin ? origin[2 * pb] : 0.0, oy = origin ? origin[2 * pb + 1] : 0.0;
    const int per = (n + 63) / 64;
    const int i0 = lane * per < n ? lane * per : n, i1 = i0 + per < n ? i0 + per : n;
    if (layout == 0) {
        for (int i = i0; i < i1; ++i) {
            out[4 * i] = ox + (double)src[4 * i];
            out[4 * i + 1] = oy + (double)src[4 * i + 1];
            out[4 * i + 2] = (double)src[4 * i + 2];
            out[4 * i + 3] = (double)src[4 * i + 3];
        }
        return;
    }
    double sx = 0.0, sy = 0.0;
    for (int i = i0; i < i1; ++i) { sx += (double)src[4 * i]; sy += (double)src[4 * i + 1]; }
    // inclusive scan of the slice sums over the wave (Hillis-Steele on shuffles), then exclusive = inclusive - own
    double ix = sx, iy = sy;
    for (int d = 1; d < 64; d <<= 1) {
        const double ux = __shfl(ix, lane >= d ? lane - d : lane), uy = __shfl(iy, lane >= d ? lane - d : lane);
        if (lane >= d) { ix += ux; iy += uy; }
    }
    const double tx = __shfl(ix, 63), ty = __shfl(iy, 63);          // closure defect of the ring
    const double cx = tx / (double)n, cy = ty / (double)n;
    double x = ox + (ix - sx) - cx * (double)i0, y = oy + (iy - sy) - cy * (double)i0;
    for (int i = i0; i < i1; ++i) {
        out[4 * i] = x;
        out[4 * i + 1] = y;
        out[4 * i + 2] = (double)src[4 * i + 2];
        out[4 * i + 3] = (double)src[4 * i + 3];
        x += (double)src[4 * i] - cx;
        y += (double)src[4 * i + 1] - cy;
    }
}

