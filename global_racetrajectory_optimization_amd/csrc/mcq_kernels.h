// mcq_kernels.h -- hand-written HIP (gfx950 / CDNA4) kernels of the minimum-curvature raceline QP engine.
//
// One workgroup == one problem (one closed reference track); the batch is the grid.  Everything is fp64.
// The maths follows SURVEY.md App. A (restating tph.opt_min_curv, call sites [REF main_globaltraj.py:264-271,
// 344-350]); DESIGN.md sections 3-5 derive the structure-exploiting form used here:
//
//   K1 mcq_assemble_kernel : [x,y,w_r,w_l] rows, normals, spline scalings  ->  box bounds, the pivots of the closed cubic-spline
//                            system (a cyclic tridiagonal matrix T: periodic pivot recurrences, mcq_tri.inc), x'', y'' (two solves
//                            with T), x', y', the curvature pre-factor, k_ref.  No matrix is formed: E = a T^-1 R Nx + b T^-1 R Ny.
//   K3 mcq_solve_kernel    : Mehrotra predictor-corrector interior point on the box QP -> active-set identification ->
//                            block-pivoting active-set iterations on the identified vertex (exact KKT point, like the
//                            Goldfarb-Idnani solver the reference uses returns) -> fp64 residual refinement through
//                            E -> curvature-row check -> opt_min_curv's curvature-error post-check.  Linear algebra: the
//                            saddle-point elimination of mcq_kkt.inc (5 x 5 blocks per waypoint; H = E'E is never formed),
//                            E and E' applied through T (mcq_tri.inc); shortest path: scalar cyclic tridiagonal (mcq_tri.inc).
//
// Global-memory workspace per problem (doubles):
//   L        [nmax][MCQ_LLD]  records of the saddle-point elimination (mcq_kkt.inc: K_AD, K_AY, K_AS, K_YV -- 65 doubles per waypoint)
//                             and three scratch vectors of the long-ring tridiagonal route (K_TS)
//   vectors  [MCQ_NVEC][nmax] see the enum below
//   Z        [nmax] + [MCQ_KMAX][MCQ_KMAX]   curvature-row path: one scratch vector, the Schur complement of the active rows
//   state    [nmax] bytes     working set
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mcq.h"

#define MCQ_LLD 72                 /* doubles per waypoint of the L slab */
#define MCQ_NVEC 32
#define MCQ_TRI_MAXN 2048         /* rings up to this length run the tridiagonal sweeps of mcq_tri.inc in LDS (eight waypoints per thread); longer ones on workspace vectors */
#define MCQ_KMAX 120               /* active curvature rows the Schur-complement path of the active-set phase holds */
#define MCQ_KBIG 512               /* ... and of the overflow path (round 3): a problem with more of them claims one of the handle's slots, */
#define MCQ_KBIG_SLOT ((size_t)MCQ_KBIG * MCQ_KBIG + 6 * (size_t)MCQ_KBIG)   /* doubles per slot: Schur matrix, three vectors, the index / sign / pivot lists */
#define MCQ_KBIG_SLOTS 8           /* slots per handle (17 MB): problems of one launch that can take the overflow path */
#define MCQ_ZLD(nmax) ((size_t)(nmax) + (size_t)MCQ_KMAX * MCQ_KMAX)   /* doubles of the curvature-row scratch per problem: one vector + the Schur matrix */

struct McqDims {
    int n;
};
__host__ __device__ inline McqDims mcq_dims(int n)
{
    McqDims d;
    d.n = n;
    return d;
}

// Pointers into HBM carry the global address space in their type so that device functions that are not inlined still
// compile to global_load / global_store (a generic pointer would become FLAT and serialise against LDS traffic).
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) signed char gschar;
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(1))) int gint;
typedef __attribute__((address_space(1))) mcq_info ginfo;

// Device workspace of ONE problem (pointers into the handle's slabs).
struct McqWork {
    const gdouble* ref;   // [n][4]
    const gdouble* nv;    // [n][2]
    const gdouble* sc;    // [n] or nullptr
    gdouble* L;           // also scratch for the T^-1 rows during assembly
    gdouble* vec;         // MCQ_NVEC vectors of length nmax, see enum below
    gschar* state;        // [n] 0 free, -1 at lower bound, +1 at upper bound, 2 fixed (lo == hi)
    gdouble* Z;           // curvature-row path: [nmax] scratch vector, then the [MCQ_KMAX][MCQ_KMAX] Schur matrix / its LU factors
    gdouble* alpha;       // [n] output
    gdouble* curv_err;    // [1]
    gint* status;         // [1]
    ginfo* info;          // [1] or nullptr
};

enum {
    V_XP = 0, V_YP, V_CP, V_KREF, V_XPP, V_YPP, V_F, V_LO, V_HI, V_X, V_G, V_ZL, V_ZU, V_SIG, V_RHS, V_DXA, V_T0, V_T1,
    V_T2, V_T3, V_TL, V_TU, V_YL, V_YU, V_SK, V_EDA, V_Q, V_NX, V_NY, V_SC,   /* unit normals and spline scalings as used (given or derived) */
    V_IDL, V_TUC        /* saddle-point core: 1 / pivot and super-diagonal of the spline matrix T (mcq_tri.inc) */
};

struct McqBatch {
    int pb0;        // slice launches (round 6, mcq_solve_host): workgroup w of the launch works on problem pb0 + w of the batch's arrays
    int batch;
    int n;          // uniform-n path (n_list == nullptr) or nmax
    int nmax;
    const int* n_list;      // per-problem n (device) or nullptr
    const double* ref;      // [batch][nmax][4]
    const double* nv;       // [batch][nmax][2], or nullptr: normals AND scalings are derived from the closed distance-scaled
                            // spline through the reference line (what prep_track does with calc_splines), `sc` is ignored
    const double* sc;       // [batch][nmax] or nullptr (unit scalings)
    double* nv_out;         // optional outputs of the assembly: normals [batch][nmax][2], scalings [batch][nmax]
    double* sc_out;
    int prep_only;          // assembly kernel stops after the spline quantities (mcq_prep_device)
    double* L; double* vec; double* Z;
    signed char* state;
    double* alpha;          // [batch][nmax]
    double* curv_err; int* status; mcq_info* info;
    double kappa_bound, w_veh;
    const double* kappa_bound_list;  // per-problem overrides (device) or nullptr
    const double* w_veh_list;
    int max_ipm_iter, max_as_iter, refine_steps, check_kappa;
    const signed char* warm; // [batch][nmax] working set to start the exchange from (IQP passes 2+), or nullptr (cold: interior point)
    int poison_lds;         // MCQ_POISON=1 (debugging aid): the solver kernel starts from an LDS full of NaNs, like the workspaces
    double* kbig;           // overflow slots of the curvature-row working set (MCQ_KBIG_SLOT doubles each), kbig_slots of them;
    int* slot_flags;        // [MCQ_SLOT_FLAGS: MCQ_KBIG_SLOTS overflow slots, MCQ_GI_FULL_MAX full, MCQ_GI_SLOTS_MAX small Goldfarb-Idnani slots] 0 = free, 1 = taken: a workgroup claims a slot by compare-and-swap and releases it when it is
                            // done (zero whenever no launch is in flight: nothing for the host to reset between launches)
    int kbig_slots;
    int objective;          // MCQ_OBJ_*: shortest path = H (a cyclic tridiagonal: two vectors) and f written directly by
                            // mcq_assemble_sp_kernel, the gradient is H x + f; no curvature rows, no curvature-error post-check
    int algorithm;          // MCQ_ALG_GI: interior point and block pivoting are skipped, every problem goes through the Goldfarb-Idnani path
    double* gi;             // FULL slots of the Goldfarb-Idnani path (mcq_gi.inc), MCQ_GI_SLOT_DOUBLES(nmax, gi_qcap) doubles each: a workgroup whose
    int gi_slots, gi_qcap;  // problem needs the path claims one (and waits for one if all are taken); gi_qcap = constraints a working set can
                            // hold (= nmax: no more can be independent)
    double* gis;            // SMALL slots (round 6; MCQ_ALG_GI: one per resident workgroup), working sets of up to gis_qcap < nmax constraints --
    int gis_slots, gis_qcap; // what the observed working sets need; a problem that outgrows its small slot moves into a full one in place (gi_grow).
                            // Flags: slot_flags[kbig_slots + MCQ_GI_FULL_MAX + s]
};

__global__ void mcq_assemble_kernel(McqBatch B);
__global__ void mcq_assemble_sp_kernel(McqBatch B);
__global__ void mcq_solve_kernel(McqBatch B);      // saddle-point elimination (mcq_kkt.inc) / scalar cyclic tridiagonal (shortest path, mcq_tri.inc); two workgroups per CU
/* Goldfarb-Idnani dual active-set path (mcq_gi.inc): inside mcq_solve_kernel, for whatever its interior point + block pivoting did not
 * settle (iteration cap, working set beyond its arrays, ...) -- solved again from scratch by quadprog's algorithm, in an HBM slot of the handle */
#define MCQ_GI_SLOTS 8
#define MCQ_GI_FULL_MAX 512        /* full slots (working sets of up to nmax constraints) a handle holds at most (the default byte cap, 16 GB, gives 256 at nmax = 2000; $MCQ_GI_BYTES moves it) */
#define MCQ_GI_SLOTS_MAX 512       /* small slots when every problem takes the path (mcq_opts.algorithm = MCQ_ALG_GI): one per resident workgroup */
#define MCQ_SLOT_FLAGS (MCQ_KBIG_SLOTS + MCQ_GI_FULL_MAX + MCQ_GI_SLOTS_MAX)
#define MCQ_GI_SLOT_FULL 100       /* gi_solve: the working set has outgrown the slot (internal: gi_rescue moves the problem to a full slot) */
#define MCQ_GI_SLOT_DOUBLES(nm, qcap) ((size_t)(qcap) * (size_t)(nm) + (size_t)(qcap) * (size_t)(qcap) + 9 * (size_t)(qcap) + 8)

/* tph.check_normals_crossing [REF helper_funcs_glob/src/prep_track.py:57-59], one workgroup per track: crossing_out [batch] =
 * 1 / 0, or -1 where tph raises (horizon >= n) */
__global__ void mcq_normals_crossing_kernel(int nmax, const int* n_list, const double* ref_all, const double* nv_all,
                                            int horizon, int* crossing_out);

/* fp32 boundary (BASELINE config 5): float <-> double streaming conversions around the fp64 engine */
__global__ void mcq_widen_kernel(const float* src, double* dst, size_t count);
__global__ void mcq_narrow_kernel(const double* src, float* dst, size_t count);
/* float rows -> fp64 rows [x, y, w_r, w_l]: layout 0 absolute coordinates, 1 ring increments (include/mcq.h MCQ_F32_*); one wave per track */
__global__ void mcq_widen_rows_kernel(const float* rows, const double* origin, double* dst, int n, int layout);

size_t mcq_solve_lds_bytes();   /* static LDS of the solver kernel (reporting only) */

/* ---- IQP glue on the device: raceline = refline + alpha * normal, closed spline through it
 *      (unit scalings), arclength re-sampling at ~stepsize, track widths carried over, normals of the re-sampled ring.
 *      One workgroup per track; mirrors tph.create_raceline / interp_track_widths / calc_splines(use_dist_scaling=False)
 *      as iqp_handler chains them. ---- */
struct McqRelin {
    int batch, nmax;
    const int* n_in;        // [batch] waypoints of each track
    const double* ref_in;   // [batch][nmax][4]
    const double* nv_in;    // [batch][nmax][2]
    const double* alpha;    // [batch][nmax]
    const int* live;        // [batch] (0: leave this track alone) or nullptr
    double alpha_scale;     // damping of the early IQP iterations (iter / iters_min), 1 afterwards
    double stepsize;        // stepsize_interp
    double* ref_out;        // [batch][nmax][4]
    double* nv_out;         // [batch][nmax][2]
    int* n_out;             // [batch]
    int* status;            // [batch]: MCQ_OK, or MCQ_BAD_INPUT if the re-sampled ring has < 3 or > nmax points
    double* vec;            // workspace, [batch][MCQ_NVEC][nmax] (the solver's vector slab)
    const signed char* state_in;   // [batch][nmax] working set the solver left for these tracks (or nullptr)
    signed char* state_out;        // [batch][nmax] the same carried to the re-sampled rings (warm start of the next pass)
};
__global__ void mcq_relinearise_kernel(McqRelin R);

/* ---- the bookkeeping of tph.iqp_handler between the QP pass and the glue, on the device (one thread per track):
 *      phase 0 (after the QP pass of round `round`): record the state of every live track (it is the final one if the track
 *              stops here), stop it if its QP failed or if round >= iters_min and curv_error_max <= curv_allowed;
 *      phase 1 (after the glue): a live track whose re-sampled ring did not fit stops with MCQ_BAD_INPUT; tracks that stopped get
 *              n_next = 0 (the next QP pass skips them).  live_count [1] = tracks still iterating (zeroed before phase 0). ---- */
struct McqIqpStep {
    int batch, phase, round, iters_min, cur;
    double curv_allowed;
    const double* curv;         // [batch] curvature error of the pass just solved
    const int* status;          // [batch] its status
    const int* relin_status;    // [batch] status of the glue (phase 1)
    int* live;                  // [batch]
    const int* n_ring;          // [batch] ring sizes solved in this round
    int* n_next;                // [batch] ring sizes of the next round (phase 1: zeroed for stopped tracks)
    int* final_n; int* final_buf; double* final_curv; int* final_status; int* final_rounds;
    double* curv_trace;         // [batch][MCQ_IQP_TRACE] or nullptr
    int* live_count;
};
__global__ void mcq_iqp_step_kernel(McqIqpStep S);

/* ---- the first `rounds` rounds of iqp_handler as ONE launch, a workgroup per track (mcq_kernels.hip: mcq_iqp_rounds_kernel).  B: the QP pass's
 *      launch parameters (outputs, options, workspace; its ref / nv / n_list / sc / warm are not read); n_set / ref_set / nv_set: the two ring
 *      buffers (round r reads set (r - 1) & 1 and writes the other); sc: the first round's scalings or nullptr; warm: the carried working sets
 *      (= R.state_out) or nullptr (cold passes); R: the glue's parameters (its n_in / ref_in / nv_in / *_out / alpha_scale are not read);
 *      S: the bookkeeping's (phase / round / cur / n_ring / n_next are not read; live_count: zero on entry, tracks still iterating on return). ---- */
struct McqIqpRounds {
    McqBatch B;
    McqRelin R;
    McqIqpStep S;
    int* n_set[2];
    double* ref_set[2];
    double* nv_set[2];
    const double* sc;
    const signed char* warm;
    int rounds;
};
__global__ void mcq_iqp_rounds_kernel(McqIqpRounds F);

/* ---- raceline at the output resolution + heading / curvature (what main_globaltraj.py runs between the QP and the velocity
 *      profile [REF main_globaltraj.py:371-387]: tph.create_raceline + tph.calc_head_curv_an).  One workgroup per track. ---- */
struct McqRace {
    int batch, nmax, mmax;
    const int* n_in;        // [batch] waypoints of each track, or nullptr (all nmax)
    const double* ref;      // [batch][nmax][4]
    const double* nv;       // [batch][nmax][2]
    const double* alpha;    // [batch][nmax]
    double stepsize;        // stepsize_interp_after_opt
    double* xy_out;         // [batch][mmax][2] raceline_interp, or nullptr
    double* psi_out;        // [batch][mmax] heading (0 = north), or nullptr
    double* kappa_out;      // [batch][mmax]
    double* el_out;         // [batch][mmax] element lengths (closed: m of them)
    int* m_out;             // [batch] points of the interpolated raceline
    int* status;            // [batch] MCQ_OK, or MCQ_BAD_INPUT (n < 3, or the raceline needs more than mmax points)
    double* vec;            // workspace, [batch][MCQ_NVEC][nmax] (the solver's vector slab)
};
__global__ void mcq_raceline_kernel(McqRace Q);

/* ---- ggv velocity profile + lap time of many (track, vehicle) variants: the forward /
 *      backward quasi-steady-state sweeps of tph.calc_vel_profile (closed track, global ggv) followed by
 *      tph.calc_ax_profile / calc_t_profile, one thread per variant. ---- */
struct McqVel {
    int batch, n, nmax;
    const int* n_of_track;   // [tracks] valid entries of each kappa / el row, or nullptr (all rows: n)
    const int* track_of;     // [batch] row of kappa / el a variant uses, or nullptr (row = variant)
    const double* kappa;     // [tracks][nmax]
    const double* el;        // [tracks][nmax] element lengths (closed: n of them)
    const double* ggv;       // [batch][ng][3]  (v, ax_max, ay_max)
    int ng;
    const double* axm;       // [batch][nam][2] (v, ax_max of the machines)
    int nam;
    const double* drag;      // [batch] drag coefficient
    const double* mass;      // [batch]
    const double* vmax;      // [batch]
    double dyn_exp;
    const double* mu;        // [tracks][nmax] friction coefficient per waypoint, or nullptr (1 everywhere)
    int filt_window;         // odd width of the closed moving-average filter on the finished profile (tph.conv_filt); <= 1: none
    double* scratch;         // [2 nmax][batch]: the lap-doubled profile, variant-minor (coalesced across threads)
    double* vx_out;          // [batch][nmax]
    double* lap_time;        // [batch]
};
__global__ void mcq_vel_profile_kernel(McqVel V);
