/*
 * mcq.h -- C ABI of the MI355X-native minimum-curvature raceline QP engine (libmcq.so).
 *
 * Drop-in boundary.  The reference has NO native interface for this path: it calls the
 * third-party Python functions
 *     trajectory_planning_helpers.opt_min_curv.opt_min_curv   [REF main_globaltraj.py:264-271, 344-350]
 *     trajectory_planning_helpers.iqp_handler.iqp_handler     [REF main_globaltraj.py:273-284]
 * which end in quadprog.solve_qp (C).  The entry points below are what a ctypes binding inside those two Python
 * functions binds instead (INTEGRATION.md shows the stub).  Plain pointers and sizes only; no torch / numpy types.
 *
 * One "problem" = one closed reference track:
 *     reftrack  [n][4] row-major double  = [x_m, y_m, w_tr_right_m, w_tr_left_m]   (producer [REF prep_track.py:39-45,104])
 *     normvec   [n][2] row-major double  = unit normals pointing right             (producer [REF prep_track.py:50-51])
 *     scaling   [n]    double            = s_i = l_i / l_{i+1}, the only information opt_min_curv needs from the
 *                                          dense 4n x 4n matrix `A` the reference passes [REF main_globaltraj.py:267]
 *                                          (s_i = -A[4i+2][4i+5], s_{n-1} = A[4n-2][1]); NULL => all ones
 *                                          (calc_splines(use_dist_scaling=False), the iqp_handler re-spline).
 * Result per problem: alpha[n] (lateral shift along the normal, metres; consumer [REF main_globaltraj.py:371-376]),
 * curv_error_max (the opt_min_curv post-check that iqp_handler terminates on), status.
 *
 * The QP solved is exactly the one tph hands to quadprog:
 *     minimise   1/2 a'Ha + f'a,   H = E'E,  f = MCQ_F_SCALE * E' k_ref
 *     subject to -(w_l - w_veh/2) <= a <= (w_r - w_veh/2),     |k_ref + E a| <= kappa_bound
 *
 * Ownership: caller owns every buffer passed in; the library copies to / from device memory it owns inside the
 * handle and retains no caller pointer after a call returns.  Threading: one host thread per handle at a time.
 * Errors: functions return 0 on success or a negative MCQ_E_* code (mcq_last_error() gives text); per-problem
 * outcomes are in status_out[].  The library never calls exit().
 */
#ifndef MCQ_H
#define MCQ_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCQ_F_SCALE 2.0 /* the factor-2 quirk of tph's quadprog call -- reproduced, not fixed */

/* per-problem status (status_out[]) */
enum {
    MCQ_OK = 0,
    MCQ_INFEASIBLE = 1,      /* w_r + w_l < w_veh somewhere  -> tph raises RuntimeError("Problem not solvable, ...") */
    MCQ_NOT_PD = 2,          /* Cholesky pivot <= 0            -> quadprog raises ValueError("matrix G is not positive definite") */
    MCQ_ITER_CAP = 3,        /* iteration cap hit -- of the Goldfarb-Idnani fallback too (20 n + 2000 steps; not observed) --, or of the interior
                               * point + block pivoting on a handle that has no Goldfarb-Idnani slot for rings this long (see MCQ_ALG_GI below) */
    MCQ_BAD_INPUT = 4,       /* n < 3, non-finite input */
    MCQ_KAPPA_INFEASIBLE = 5, /* curvature rows cannot be satisfied -> quadprog raises ValueError("constraints are inconsistent, no solution") */
    MCQ_KAPPA_ACTIVE = 6,     /* box-only optimum violates a curvature row and the curvature-row phase is disabled (check_kappa < 0).
                               * (Rounds 1-4 also returned it for more active curvature rows than the working-set arrays hold -- 120 in LDS,
                               * 512 through an overflow slot; such problems now go through the Goldfarb-Idnani path, which has no limit --
                               * round 6: ALL of them, so that a problem's route and the last bits of its result do not depend on what else is in
                               * the launch.  On a handle without a Goldfarb-Idnani slot -- rings beyond the byte cap below -- the overflow slots
                               * serve as before and this status can come back for more than 512 rows.)  Also returned by that path when
                               * check_kappa < 0 and the returned point violates a curvature row. */
    MCQ_RING_OVERFLOW = 7,    /* mcq_iqp_device / mcq_iqp_batch only: the re-sampled raceline of an IQP round needs more waypoints than
                               * the buffers hold (nmax / nmax_out) -- not an input error of the QP (that stays MCQ_BAD_INPUT) */
    MCQ_KAPPA_NO_SLOT = 8     /* returned only on a handle WITHOUT Goldfarb-Idnani slots (rings beyond the byte cap): more than eight problems of
                               * one launch with more than 120 active curvature rows each.  Everywhere else such problems are solved by the
                               * Goldfarb-Idnani path inside the same kernel, on every entry point */
};

/* library-level error codes (negative return values) */
enum {
    MCQ_E_ARG = -1,     /* NULL / out-of-range argument */
    MCQ_E_DEVICE = -2,  /* HIP runtime error (no device, launch failure, out of memory) */
    MCQ_E_TOO_LARGE = -3
};

typedef struct mcq_handle mcq_handle; /* opaque; owns device buffers + one HIP stream on one device */

typedef struct {
    int n;                  /* waypoints */
    const double* reftrack; /* [n][4] */
    const double* normvec;  /* [n][2], or NULL for every problem of a batch: derived on the device (with the scalings) */
    const double* scaling;  /* [n] or NULL */
    double kappa_bound;     /* veh_params.curvlim  [REF params/racecar.ini:49] */
    double w_veh;           /* optim_opts.width_opt [REF params/racecar.ini:72] */
} mcq_problem;

typedef struct {
    int algorithm;      /* MCQ_ALG_DEFAULT (0; any value other than MCQ_ALG_GI): interior point -> block pivoting on the identified vertex,
                         * with the Goldfarb-Idnani path below as the fallback of every problem that phase does not settle;
                         * MCQ_ALG_GI (1): EVERY problem through the engine's Goldfarb-Idnani dual active-set path (quadprog's algorithm
                         * [REF requirements.txt:3 via tph.opt_min_curv]: one constraint enters per iteration, ratio test, drops -- finite by
                         * construction; a reference mode: 8.1 k solves/s on 1024 rings of 2000 waypoints, a twelfth of the default path's rate).
                         * Memory of that path (round 6): FULL slots (2 nmax^2 doubles: a working set of up to nmax constraints) for the fallback,
                         * as many as $MCQ_GI_BYTES (default 16 GB) hold, at most 512, never more than the batch -- and NONE where one slot exceeds
                         * the cap (rings above ~32 000 waypoints) or the allocation is refused: the launch then runs without the fallback and
                         * reports what interior point + block pivoting left (MCQ_ITER_CAP ...), it does not fail; SMALL slots (working sets of up
                         * to max(128, nmax / 8) constraints, 4.6 MB at nmax = 2000) for MCQ_ALG_GI, one per resident workgroup (2.4 GB for 512),
                         * a problem that outgrows its small slot moving into a full one.  MCQ_ALG_GI without any slot to be had is MCQ_E_DEVICE.
                         * mcq_solve_host_pipelined / mcq_solve_device_stream keep MCQ_ALG_GI on ONE compute stream.  Minimum-curvature
                         * objective only.  (Until round 4 this field was `band_e`, ignored since E is applied through the spline system.) */
    int max_ipm_iter;   /* 0 => default 60 */
    int max_as_iter;    /* 0 => default 60 */
    int refine_steps;   /* fp64 residual-refinement rounds on the final active set; <0 => default 2 */
    int check_kappa;    /* curvature rows |k_ref + E a| <= kappa_bound: 0 or 1 => carried (the default: a zero-initialised mcq_opts
                         * solves the QP the reference solves); < 0 => skipped (box-only QP; a violated row is then reported as
                         * MCQ_KAPPA_ACTIVE instead of being enforced) */
    int objective;      /* MCQ_OBJ_MIN_CURV (0, default) or MCQ_OBJ_SHORTEST_PATH: the QP of tph.opt_shortest_path
                         * [REF main_globaltraj.py:286-290] on the same box rows -- H = cyclic tridiagonal
                         * (4 |n_i|^2 on the diagonal, -2 n_i.n_{i+1} beside it), f_i = 2 n_i.(2 p_i - p_{i-1} - p_{i+1}),
                         * deviations clipped at 0.001 m instead of rejected.  normvec is required; scaling, kappa_bound
                         * and the curvature rows are ignored and curv_err_out is 0. */
    int warm_start;     /* 1: start from the working sets that the preceding mcq_relinearise_device call on this handle carried
                         * over from the preceding solve of the same batch (IQP passes 2+): the exchange alone, no interior point,
                         * unless it runs out of 12 rounds (then the cold path; mcq_info.second_attempt & 2).  Ignored when no such
                         * working sets are at hand.  The result is an exact KKT vertex of the same QP either way.  Default 0. */
} mcq_opts;
#define MCQ_OBJ_MIN_CURV 0
#define MCQ_OBJ_SHORTEST_PATH 1
#define MCQ_ALG_DEFAULT 0
#define MCQ_ALG_GI 1

/* per-problem diagnostics (optional output) */
typedef struct {
    int ipm_iters;      /* interior-point iterations (one factorisation each) */
    int as_iters;       /* active-set (block pivoting) iterations (one factorisation each) */
    int n_active_box;   /* box rows active at the optimum */
    int n_active_kappa; /* curvature rows active at the optimum */
    double kappa_max;   /* max_i |k_ref_i + (E a)_i| at the returned a */
    double kkt_res;     /* max free-gradient magnitude relative to max |f| */
    /* device wall-clock (s_memrealtime, 100 MHz ticks) spent by this problem's workgroup in the phases of the solver
     * kernel: [0] factorisations, [1] solves, [2] gradients (E, E' through the spline system), [3] whole kernel (assembly included),
     * [4] / [5] the interior-point / the active-set phase, [7] curvature check, (rare) curvature-row phase, outputs;
     * [6] the same span as [3] in shader-clock cycles (s_memtime): [6] / [3] x 100 MHz = the clock the CU actually ran at */
    long long ticks[8];
    int refine_rounds;  /* fp64 refinement rounds run on the final working set (<= opts.refine_steps) */
    int second_attempt; /* bit 0: the active-set phase ran out of its first budget and the interior point was resumed to
                           mu = 1e-13 (degenerate / very ill-conditioned instance); bit 1: a warm start (mcq_opts.warm_start)
                           was abandoned for the cold path */
    int f32_factorisations;  /* of ipm_iters: factorisations whose records were stored as floats (the first interior-point
                                iterations, while the dual residual is far above what such records can resolve; round 4) */
    int gi_iters;       /* iterations (full + partial steps) of the Goldfarb-Idnani path, 0 if it did not run for this problem
                           (second_attempt bit 2 says that it ran, bit 3 that its final polish through the block-pivoting phase did not
                           settle and the Goldfarb-Idnani iterate itself is returned: feasible / optimal to its own 2e-9 m tolerances;
                           bits 4-7: the status the solver kernel had left for the problem -- why this path ran) */
} mcq_info;

int mcq_create(int device_id, mcq_handle** out);
void mcq_destroy(mcq_handle* h);
const char* mcq_last_error(void);
void mcq_default_opts(mcq_opts* o);

/* Host-buffer entry point == what opt_min_curv binds.  alpha_out is the concatenation of the per-problem alpha
 * vectors (sum of n over the batch); curv_err_out/status_out/info_out have `batch` entries (info_out may be NULL). */
int mcq_solve_batch(mcq_handle* h, const mcq_problem* probs, int batch, const mcq_opts* opts, double* alpha_out,
                    double* curv_err_out, int* status_out, mcq_info* info_out);

/* Device-resident entry points (uniform n, inputs already in HBM; used by bench.py, the IQP driver and the multi-GPU
 * shard path).  All pointers are DEVICE pointers on the handle's device; layouts as above with a leading batch axis:
 * reftrack [batch][n][4], normvec [batch][n][2], scaling [batch][n] or NULL, alpha_out [batch][n],
 * curv_err_out [batch], status_out [batch], info_out [batch] or NULL.  Asynchronous on the handle's stream;
 * mcq_sync() waits.  mcq_stream() returns the hipStream_t (as void*) so callers can record events on it. */
int mcq_solve_device(mcq_handle* h, int batch, int n, const double* reftrack, const double* normvec,
                     const double* scaling, double kappa_bound, double w_veh, const mcq_opts* opts, double* alpha_out,
                     double* curv_err_out, int* status_out, mcq_info* info_out);
/* The same with per-problem sizes: n_list [batch] (device) waypoints per problem, every array strided by nmax
 * (reftrack [batch][nmax][4], ...).  This is what a batch of IQP runs needs: N changes from pass to pass and differs
 * between tracks [REF main_globaltraj.py:273-284]. */
int mcq_solve_device_ragged(mcq_handle* h, int batch, int nmax, const int* n_list, const double* reftrack,
                            const double* normvec, const double* scaling, double kappa_bound, double w_veh,
                            const mcq_opts* opts, double* alpha_out, double* curv_err_out, int* status_out,
                            mcq_info* info_out);

/* mcq_solve_device_ragged with per-problem vehicle parameters: kappa_bound_list / w_veh_list [batch] (DEVICE arrays; either may
 * be NULL, then the scalar applies to every problem) -- what mcq_problem.{kappa_bound,w_veh} are to the host-buffer entry.  The
 * shape of a vehicle-width sweep over resident tracks [REF params/racecar.ini:49,72; BASELINE config 4]. */
int mcq_solve_device_ragged_params(mcq_handle* h, int batch, int nmax, const int* n_list, const double* reftrack,
                                   const double* normvec, const double* scaling, double kappa_bound, double w_veh,
                                   const double* kappa_bound_list, const double* w_veh_list, const mcq_opts* opts,
                                   double* alpha_out, double* curv_err_out, int* status_out, mcq_info* info_out);

/* fp32 at the boundary (BASELINE config 5: 65536 tracks, alpha collected with one all-gather): as mcq_solve_device, with
 * the tracks stored as float in HBM (reftrack [batch][n][4], normvec [batch][n][2] or NULL, scaling [batch][n] or NULL) and
 * alpha_out [batch][n] written as float.  The arithmetic in between is the fp64 engine, unchanged -- cond(H) = 1e9..1e12
 * leaves no room for fp32 factors -- so the result is the exact QP solution OF THE ROUNDED INPUTS, rounded once more on
 * the way out (|alpha| < 2^3 m => 2.4e-7 m); how far the rounded inputs move the solution is a property of the QP, not of
 * the engine (measured in tests/test_gpu_parity.py).  With normvec == NULL the normals and scalings are derived in fp64 from
 * the float x, y (preferred: float normals are unit vectors only to 6e-8).  curv_err_out stays double. */
int mcq_solve_device_f32(mcq_handle* h, int batch, int n, const float* reftrack, const float* normvec,
                         const float* scaling, double kappa_bound, double w_veh, const mcq_opts* opts, float* alpha_out,
                         double* curv_err_out, int* status_out, mcq_info* info_out);

/* fp32 rows in the layout that keeps the QP's accuracy (round 3).  The QP depends on the coordinates only through differences
 * of neighbouring waypoints (x', x'' of the closed spline), so rounding ABSOLUTE coordinates to float (ulp 1.2e-4 m at |x| = 1.9 km)
 * throws away what the 3 m steps carry: measured 2e-3 m on alpha at N = 2000.  MCQ_F32_INCREMENTS stores the ring as float
 * increments -- row i = [x_{i+1} - x_i, y_{i+1} - y_i, w_tr_right_i, w_tr_left_i], row n-1 closing the ring -- plus an optional fp64
 * origin per track (alpha does not depend on it): the device rebuilds x, y by an fp64 running sum whose closure defect (the float
 * increments do not sum to zero exactly: ~1e-5 m) is spread evenly over the n increments.  Measured: |alpha - alpha(fp64 rows)|
 * <= 5e-6 m at N = 2000, the same as rounding the two WIDTH columns alone; stated tolerance of BASELINE config 5: 1e-4 m
 * (tests/test_gpu_parity.py::test_fp32_boundary_full_size).  Normals and scalings are always derived on the device in fp64.
 * layout = MCQ_F32_ABSOLUTE takes rows [x, y, w_r, w_l] like mcq_solve_device_f32 (origin added if given). */
#define MCQ_F32_ABSOLUTE 0
#define MCQ_F32_INCREMENTS 1
/* device-resident: reftrack [batch][n][4] float, origin [batch][2] double or NULL, alpha_out [batch][n] float (DEVICE pointers) */
int mcq_solve_device_f32_rows(mcq_handle* h, int batch, int n, int layout, const float* reftrack, const double* origin,
                              double kappa_bound, double w_veh, const mcq_opts* opts, float* alpha_out, double* curv_err_out,
                              int* status_out, mcq_info* info_out);
/* host buffers (SURVEY.md section 8b's `mcq_solve_batch_f32`): the same for a uniform batch in HOST memory -- float rows in, float
 * alpha out, half the PCIe bytes of mcq_solve_host; curv_err_out / status_out / info_out (may be NULL) in host memory.  Blocking. */
int mcq_solve_batch_f32(mcq_handle* h, int batch, int n, int layout, const float* reftrack, const double* origin,
                        double kappa_bound, double w_veh, const mcq_opts* opts, float* alpha_out, double* curv_err_out,
                        int* status_out, mcq_info* info_out);

/* The front half of prep_track on the device [REF helper_funcs_glob/src/prep_track.py:48-51]: unit normals (pointing
 * right) and spline scalings s_i = l_i / l_{i+1} of the closed distance-scaled cubic spline through the reference line --
 * what tph.calc_splines(path) returns as normvec_normalized and encodes in its matrix.  reftrack [batch][nmax][4],
 * n_list [batch] or NULL (all nmax); outputs normvec_out [batch][nmax][2] and / or scaling_out [batch][nmax] (either may
 * be NULL); status_out [batch].  The solve entry points derive the same quantities themselves when they are called with
 * normvec == NULL (then `scaling` is ignored).  Asynchronous on the handle's stream. */
int mcq_prep_device(mcq_handle* h, int batch, int nmax, const int* n_list, const double* reftrack, double* normvec_out,
                    double* scaling_out, int* status_out);

/* The check prep_track runs on the normals [REF helper_funcs_glob/src/prep_track.py:57-59] -- tph.check_normals_crossing(track,
 * normvec_normalized, horizon = 10): do the normal segments [p - w_left n, p + w_right n] of two waypoints at most `horizon`
 * apart intersect?  reftrack [batch][nmax][4], normvec [batch][nmax][2], n_list [batch] or NULL (all nmax);
 * crossing_out [batch] (device) = 1 / 0, or -1 where tph raises RuntimeError (horizon >= n).  Asynchronous on the handle's
 * stream. */
int mcq_normals_crossing_device(mcq_handle* h, int batch, int nmax, const int* n_list, const double* reftrack,
                                const double* normvec, int horizon, int* crossing_out);

/* Device-side glue of tph.iqp_handler between two passes (what upstream does on the host with two dense 4N x 4N spline
 * solves per pass): raceline = refline + alpha_scale * alpha * normal; closed spline through it (unit scalings);
 * arclength re-sampling at ~stepsize (tph.create_raceline); track widths shifted by -/+ alpha and carried over linearly
 * (tph.interp_track_widths); unit normals of the closed spline through the re-sampled ring
 * (tph.calc_splines(use_dist_scaling=False)).  All pointers DEVICE pointers, arrays strided by nmax; `live` [batch] or
 * NULL selects the tracks to process; n_out [batch] receives the new waypoint counts; status_out [batch] MCQ_OK or
 * MCQ_BAD_INPUT (new ring would have < 3 or > nmax points; the IQP loop reports that as MCQ_RING_OVERFLOW).  Input and output buffers must differ.  Asynchronous on
 * the handle's stream.  Side effect inside the handle: the working sets the preceding solve left for these tracks are carried to
 * the re-sampled rings (new point -> nearer end of its old segment), for mcq_opts.warm_start of the next pass. */
int mcq_relinearise_device(mcq_handle* h, int batch, int nmax, const int* n_in, const double* reftrack_in,
                           const double* normvec_in, const double* alpha, const int* live, double alpha_scale,
                           double stepsize, double* reftrack_out, double* normvec_out, int* n_out, int* status_out);
/* ggv velocity profile and lap time of a batch of (track, vehicle) variants -- what the lap-time-matrix sweep of the
 * reference runs per cell [REF main_globaltraj.py:400-422, 460-493]: tph.calc_vel_profile (closed track, global ggv, no
 * filter, mu = 1 -- mcq_vel_profile_device_opts below takes both; every step of upstream's solver: fixed-point lateral limit over all ggv rows, sweeps gated on the acceleration-phase
 * starts and on v_max, one look-ahead round in the backward sweep) followed by calc_ax_profile / calc_t_profile (lap time as
 * the sum of 2 l / (v_a + v_b)).  One device thread per variant.  A variant whose ggv or machine table ends below its v_max
 * (tph raises RuntimeError) gets lap_time NaN and a vx_out row of NaNs (never stale buffer contents).  All pointers DEVICE pointers:
 * kappa / el_lengths [tracks][nmax] (n valid entries each), track_of [batch] (row used by a variant) or NULL (row =
 * variant), ggv [batch][n_ggv][3] (v, ax_max, ay_max), ax_max_machines [batch][n_machines][2], drag_coeff / m_veh / v_max
 * [batch]; outputs vx_out [batch][nmax], lap_time_out [batch].  Asynchronous on the handle's stream. */
int mcq_vel_profile_device(mcq_handle* h, int batch, int n, int nmax, const int* track_of, const double* kappa,
                           const double* el_lengths, const double* ggv, int n_ggv, const double* ax_max_machines,
                           int n_machines, const double* drag_coeff, const double* m_veh, const double* v_max,
                           double dyn_model_exp, double* vx_out, double* lap_time_out);

/* The same over tracks of different lengths: n_of_track [tracks] (device) = valid entries of each kappa / el_lengths row (rows
 * strided by nmax); a variant whose row has fewer than 2 or more than nmax entries gets lap_time NaN.  This is the shape of
 * BASELINE config 4's sweep: (reference track x vehicle width) racelines from mcq_raceline_device, each shared by a grid of
 * ggv / top-speed variants. */
int mcq_vel_profile_device_ragged(mcq_handle* h, int batch, int nmax, const int* n_of_track, const int* track_of,
                                  const double* kappa, const double* el_lengths, const double* ggv, int n_ggv,
                                  const double* ax_max_machines, int n_machines, const double* drag_coeff,
                                  const double* m_veh, const double* v_max, double dyn_model_exp, double* vx_out,
                                  double* lap_time_out);

/* The same two entries with the options of tph.calc_vel_profile that main_globaltraj.py reads from its parameter file
 * [REF main_globaltraj.py:400-410, params/racecar.ini:53-57: vel_calc_opts] or leaves at their defaults: dyn_model_exp; filt_window --
 * tph.conv_filt's closed moving average of that (odd) width over the finished profile, lap time from the filtered profile (0 or 1:
 * none; an even width, where tph raises RuntimeError, or one wider than the ring gives lap_time NaN); mu -- a friction coefficient per
 * waypoint, DEVICE pointer [tracks][nmax] like kappa, or NULL for 1 everywhere (tph's `mu` argument; the reference passes none).
 * n_of_track NULL: every row has n entries (mcq_vel_profile_device); otherwise n is ignored (.._ragged). */
typedef struct mcq_vel_opts {
    double dyn_model_exp;
    int filt_window;
    int reserved_;
    const double* mu;
} mcq_vel_opts;
int mcq_vel_profile_device_opts(mcq_handle* h, int batch, int n, int nmax, const int* n_of_track, const int* track_of,
                                const double* kappa, const double* el_lengths, const double* ggv, int n_ggv,
                                const double* ax_max_machines, int n_machines, const double* drag_coeff, const double* m_veh,
                                const double* v_max, const mcq_vel_opts* opts, double* vx_out, double* lap_time_out);

/* What main_globaltraj.py runs between the QP and the velocity profile [REF main_globaltraj.py:371-387], batched on the device:
 * tph.create_raceline (raceline = refline + alpha * normal, closed cubic spline through it with unit scalings, re-sampled at
 * ~stepsize = stepsize_interp_after_opt) and tph.calc_head_curv_an (heading and curvature from the spline's derivatives).
 * Inputs strided by nmax as in mcq_solve_device_ragged (n_in [batch] or NULL: all nmax); outputs strided by mmax:
 * raceline_out [batch][mmax][2] (or NULL), psi_out [batch][mmax] (heading, 0 = north, [-pi, pi); or NULL), kappa_out and
 * el_lengths_out [batch][mmax] -- the two inputs of mcq_vel_profile_device(_ragged) --, m_out [batch] points per raceline,
 * status_out [batch] MCQ_OK or MCQ_BAD_INPUT (n < 3, or more than mmax points needed).  Asynchronous on the handle's stream. */
int mcq_raceline_device(mcq_handle* h, int batch, int nmax, const int* n_in, const double* reftrack, const double* normvec,
                        const double* alpha, double stepsize, int mmax, double* raceline_out, double* psi_out,
                        double* kappa_out, double* el_lengths_out, int* m_out, int* status_out);

/* Host-buffer entry for a UNIFORM batch (every track n waypoints): reftrack [batch][n][4], normvec [batch][n][2] or NULL,
 * scaling [batch][n] or NULL in host memory, results to host memory.  One asynchronous copy per array straight from / to the
 * caller's buffers -- no packing pass; buffers from mcq_host_alloc (pinned) are copied at PCIe speed, pageable ones go through
 * the runtime's staging.  This is the wall SURVEY.md section 8d defines the metric on ("inputs resident in host pinned memory
 * -> alpha resident in host memory"); bench.py reports it next to the device-resident rate.  Blocking.  Round 6: a batch of 512 or more
 * (minimum-curvature objective, default algorithm) goes in two SLICES, one per compute stream of the handle -- the second slice's upload and the
 * first slice's download overlap the kernels (11.8 ms where the one launch took 13.0 for 1024 x N = 2000; four slices: 11.5, but then a heavy-tailed
 * batch waits for its slowest problems slice by slice);
 * results bitwise those of the one launch; mcq_last_timing is not valid after such a call.  mcq_solve_batch does the same behind its packing
 * threads.  Knobs (environment): MCQ_HOST_ONE_LAUNCH=1, MCQ_HOST_SLICES (2 .. 8), MCQ_HOST_SLICE_MIN. */
int mcq_solve_host(mcq_handle* h, int batch, int n, const double* reftrack, const double* normvec, const double* scaling,
                   double kappa_bound, double w_veh, const mcq_opts* opts, double* alpha_out, double* curv_err_out,
                   int* status_out, mcq_info* info_out);

/* A STREAM of uniform batches from / to host memory with the PCIe hidden behind the kernels (SURVEY.md section 8d defines the
 * metric "inputs resident in host pinned memory -> alpha resident in host memory"): step k's kernels run while step k+1's rows are
 * uploaded and step k-1's results are downloaded, on two copy streams and two sets of device staging buffers -- and, since round 5, on TWO
 * COMPUTE STREAMS with a workspace each (a second 1.8 MB per problem while the entry is in use): consecutive steps are independent, so the
 * kernels of step k+1 start on the compute units step k's launch has already left instead of waiting for its slowest problems.
 * curv_err_out / status_out may be pageable (they pass through pinned staging of the handle and are filled when the call returns); alpha_out and
 * the inputs should be pinned (mcq_host_alloc).  Debugging knobs (environment): MCQ_PIPE_ONE_STREAM=1 keeps every step on the handle's first
 * compute stream (round 4's behaviour, for A/B measurements), MCQ_PIPE_TRACE=1 prints when the host enqueued each step.  Arrays of `steps`
 * host pointers (pinned memory from mcq_host_alloc for full PCIe speed): reftrack[k] [batch][n][4], normvec[k] [batch][n][2] (array or
 * entries may be NULL: derived on the device), scaling[k] [batch][n] (may be NULL), alpha_out[k] [batch][n], curv_err_out[k] [batch],
 * status_out[k] [batch].  Results of step k are bitwise those of mcq_solve_host on the same buffers.  Blocking; returns when
 * every step's results are in host memory. */
int mcq_solve_host_pipelined(mcq_handle* h, int steps, int batch, int n, const double* const* reftrack, const double* const* normvec,
                             const double* const* scaling, double kappa_bound, double w_veh, const mcq_opts* opts,
                             double* const* alpha_out, double* const* curv_err_out, int* const* status_out);

/* The same for batches already RESIDENT in device memory (round 5): arrays of `steps` DEVICE pointers, layouts as for mcq_solve_device; normvec / scaling
 * arrays or entries may be NULL.  The launches of consecutive steps alternate between the handle's two compute streams (a workspace each), so a
 * step's slowest problems finish while the next step's workgroups already fill the compute units the others have left: 113 k solves/s where
 * launch-by-launch calls of mcq_solve_device give 98 k (1024 rings of 2000 waypoints).  The steps must be independent of each other (distinct
 * output buffers; an input may repeat).  Asynchronous: ordered behind what the handle's stream holds when it is called; mcq_sync (or any later
 * call on the handle) waits for every step.  Results of step k are bitwise those of mcq_solve_device on the same buffers. */
int mcq_solve_device_stream(mcq_handle* h, int steps, int batch, int n, const double* const* reftrack, const double* const* normvec,
                            const double* const* scaling, double kappa_bound, double w_veh, const mcq_opts* opts, double* const* alpha_out,
                            double* const* curv_err_out, int* const* status_out);

/* ---- tph.iqp_handler [REF main_globaltraj.py:273-284] as ONE call: the whole iterated re-linearisation of a batch of tracks.
 *
 * Every round is one batched QP pass (mcq_solve_device_ragged; passes 2+ warm-started from the working set the glue carried
 * over) followed, on the device, by the termination test of iqp_handler (round >= iters_min and curv_error_max <=
 * curv_error_allowed), the damping of the early rounds (alpha * round / iters_min) and the glue of mcq_relinearise_device for
 * the tracks that go on.  The host enqueues the first iters_min rounds without looking and then reads ONE int per round (how
 * many tracks are still iterating).  A track whose QP fails keeps that status and stops; the others are not affected.
 * Those first rounds are ONE launch in which every workgroup takes its track through the rounds on its own (a track's passes
 * depend on its own previous pass only: launched round by round, every round would wait for the batch's slowest track);
 * results bitwise those of one launch per round, which is what runs with `stats->timed`, with a round callback, or with
 * $MCQ_IQP_FUSED=0.
 *
 * mcq_iqp_device: everything resident.  reftrack_a / normvec_a [batch][nmax][*] hold the tracks on entry (n_io [batch] their
 * waypoint counts), reftrack_b / normvec_b are the second set of the double buffer; scaling [batch][nmax] (first pass) or NULL.
 * On return: alpha_out [batch][nmax] = alpha of the last pass (damped if the track ended in an early round -- only with
 * iters_min > the rounds run, as upstream), n_io = waypoint counts of the last re-linearisation, buf_out [batch] = 0 / 1: which
 * set holds a track's final reftrack / normvectors, curv_err_out / status_out / rounds_out [batch]; curv_trace_out
 * [batch][MCQ_IQP_TRACE] (optional) = curv_error_max of every round of a track (what iqp_handler prints with print_debug;
 * rounds beyond MCQ_IQP_TRACE are not recorded: rounds_out tells when the trace is truncated).  A track with n_io == 0 on entry is
 * not a track: it never runs (status MCQ_BAD_INPUT from its first, empty pass, rounds_out 1).  stats (optional, host): see mcq_iqp_stats.  Blocking (the loop needs the
 * live count). */
#define MCQ_IQP_TRACE 16
typedef struct {
    int rounds;             /* rounds run (the slowest track) */
    int qp_solves;          /* QP passes summed over the tracks */
    float solver_ms[16];    /* per round: the QP pass's launch sequence (HIP events; rounds beyond 16 are not recorded; only
                             * filled when `timed` is set on entry: costs one event synchronisation per round) */
    int fallbacks[16];      /* per round: warm starts abandoned for the cold path (needs info read-back: only when `timed`) */
    int timed;              /* in: 1 => fill solver_ms / fallbacks */
} mcq_iqp_stats;
int mcq_iqp_device(mcq_handle* h, int batch, int nmax, int* n_io, double* reftrack_a, double* normvec_a, double* reftrack_b,
                   double* normvec_b, const double* scaling, double kappa_bound, double w_veh, double stepsize_interp,
                   int iters_min, double curv_error_allowed, int max_rounds, const mcq_opts* opts, double* alpha_out,
                   int* buf_out, double* curv_err_out, int* status_out, int* rounds_out, double* curv_trace_out,
                   mcq_iqp_stats* stats);
/* The same from / to host buffers -- what iqp_handler binds.  problems [batch] as for mcq_solve_batch (normvec required; batches
 * above 8 MB are packed into the pinned staging by $MCQ_PACK_THREADS host threads -- default 8 -- in chunks whose uploads overlap
 * the packing of the next; mcq_solve_batch packs the same way);
 * outputs padded to nmax_out waypoints per track (caller's capacity for the re-sampled rings; MCQ_E_TOO_LARGE names nothing
 * partial: a track whose ring outgrows it gets status MCQ_BAD_INPUT): alpha_out [batch][nmax_out], reftrack_out
 * [batch][nmax_out][4], normvec_out [batch][nmax_out][2], n_out / status_out / rounds_out [batch], curv_err_out [batch],
 * curv_trace_out [batch][MCQ_IQP_TRACE] or NULL. */
int mcq_iqp_batch(mcq_handle* h, const mcq_problem* probs, int batch, double stepsize_interp, int iters_min,
                  double curv_error_allowed, int max_rounds, const mcq_opts* opts, int nmax_out, double* alpha_out,
                  double* reftrack_out, double* normvec_out, int* n_out, double* curv_err_out, int* status_out,
                  int* rounds_out, double* curv_trace_out, mcq_iqp_stats* stats);

/* Optional per-round callback of mcq_iqp_device / mcq_iqp_batch (what tph.iqp_handler prints per iteration with print_debug
 * [REF main_globaltraj.py:270,280]): called on the host after the QP pass of every round, before its termination test, with the
 * curvature errors of the pass and the tracks that ran it (live[k] != 0).  Costs one small device-to-host copy and a synchronisation per
 * round while set; cb == NULL removes it. */
typedef void (*mcq_iqp_round_cb)(void* user, int round, int batch, const double* curv_err, const int* live);
int mcq_iqp_set_round_callback(mcq_handle* h, mcq_iqp_round_cb cb, void* user);

/* The one thing the engine reads from the dense spline matrix the reference passes [REF main_globaltraj.py:267 `A=a_interp`;
 * helper_funcs_glob/src/prep_track.py:48-51]: the n spline scalings s_i encoded in the [4n][4n] row-major matrix of tph.calc_splines' closed-spline
 * system (s_i = -A[4i+2][4i+5], s_(n-1) = A[4n-2][1]) -> s_out [n].  check != 0 verifies that A IS that matrix -- the 12 n structural entries at
 * their places and values, the four that mirror the scalings, and not one non-zero anywhere else -- in ONE pass over the matrix on a host thread per 16 MB
 * (32 at most; $MCQ_PACK_THREADS overrides) (512 MB at n = 2000: the numpy restatement of the same check in trajectory_planning_helpers/calc_splines.py took 100-200 ms of
 * a call whose kernel takes 3; round 6).  No GPU involved, no handle.  Returns 0, or MCQ_E_ARG when A does not have the structure (the message
 * names the first offending entry). */
int mcq_les_scalings(const double* A, int n, double* s_out, int check);

/* Pinned (page-locked) host memory for callers that want their buffers copied at PCIe speed (mcq_solve_host, mcq_solve_batch,
 * mcq_copy_*). */
int mcq_host_alloc(mcq_handle* h, size_t bytes, void** out);
int mcq_host_free(mcq_handle* h, void* ptr);

/* Device memory plumbing on the handle's device and stream, for callers that keep data resident between calls (the
 * Python IQP driver) without loading a second HIP runtime into the process: allocate (zero-filled) / free / blocking
 * copies.  A reference-side binding would use these exactly where a CUDA/HIP-aware caller uses its own allocator. */
int mcq_device_alloc(mcq_handle* h, size_t bytes, void** out);
int mcq_device_free(mcq_handle* h, void* ptr);
int mcq_copy_to_device(mcq_handle* h, void* dst, const void* src, size_t bytes);
int mcq_copy_to_host(mcq_handle* h, void* dst, const void* src, size_t bytes);
int mcq_sync(mcq_handle* h);
void* mcq_stream(mcq_handle* h);

/* Timing of the last mcq_solve_device / mcq_solve_batch call, measured with HIP events on the handle's stream:
 * ms[2] the solver kernel (since round 4 the whole QP pass: assembly, solve, post-check -- and, round 5, the Goldfarb-Idnani path of the
 * problems that need it), ms[4] the whole launch sequence; ms[0] the assembly kernel of the shortest-path objective (else ~0);
 * ms[1], ms[3] are 0 (the kernels they timed until round 3 no longer exist; the slots stay for ABI compatibility). */
int mcq_last_timing(mcq_handle* h, float ms[5]);

/* A SPAN of launches timed on the device: mcq_timing_begin records an event on the handle's compute stream, mcq_timing_end records a second one,
 * waits for it and returns the milliseconds between the two and the number of solver launches the handle enqueued on that stream in between
 * (mcq_solve_device* / the QP passes of the IQP entries; not the second stream of mcq_solve_host_pipelined).  With launches enqueued back to back
 * -- no host synchronisation inside the span -- ms / launches is the average duration of a launch, launch gaps included: what bench.py's
 * `roofline.kernel_ms` is (round 5; until then it synchronised after every step to read mcq_last_timing, and a slow host showed up in `value`). */
int mcq_timing_begin(mcq_handle* h);
int mcq_timing_end(mcq_handle* h, float* ms_out, int* launches_out);

/* Bytes of device workspace the handle currently holds (for DESIGN.md / bench reporting). */
long long mcq_workspace_bytes(mcq_handle* h);

/* 1 if the last host-buffer batch of problem records (mcq_solve_batch, mcq_iqp_batch) was uploaded WITHOUT the packing pass (round 6): a
 * uniform batch -- every track n waypoints -- whose reftrack / normvec / scaling rows lie back to back in page-locked memory (mcq_host_alloc) goes
 * to the device straight from there, one strided copy per array; anything else (ragged, pageable, scattered) is packed into the handle's pinned
 * staging first (several host threads, chunk by chunk).  $MCQ_PACK_ALWAYS=1 forces the packing pass.  Same results either way. */
int mcq_last_upload_was_direct(mcq_handle* h);

/* ---- the one collective of a multi-GPU job (north_star: "a single RCCL all-gather over xGMI to collect the alpha
 *      vectors").  One process per GPU, one handle per process; independent QPs are block-partitioned over the ranks and every rank
 *      solves its shard with the entries above -- no collective inside a solve.  The gather is the engine's own: ncclAllGather of RCCL
 *      (loaded with dlopen on first use: $MCQ_RCCL_LIB, else $ROCM_PATH/lib/librccl.so.1, else /opt/rocm/lib/librccl.so.1, else
 *      librccl.so.1 -- a process that never gathers never loads it).  ORDERING, stated once: a gather runs on a COMM STREAM of the handle,
 *      ordered behind everything enqueued on the handle's compute stream at the time of the call (an event) -- so after the solve that
 *      filled `send` -- and concurrent with whatever is enqueued on the compute stream afterwards (the next solve).  Work on the compute
 *      stream is NOT ordered behind a gather; a caller waits with mcq_comm_wait or mcq_sync before it reads `recv` or overwrites `send`.
 *      (The blocking helpers mcq_copy_to_host and mcq_device_free do wait for the gathers enqueued before them, since round 5.)
 *      No torch, no second HIP runtime in the process.
 *      Reference side: nothing to replace -- the reference is a single-process script; a caller that shards its sweeps
 *      [REF main_globaltraj.py:441-505, the lap-time matrix loops] would call these three.
 *   mcq_comm_unique_id   rank 0 creates the 128-byte id and ships it to the other ranks by whatever means the launcher offers
 *                        (bench.py: one broadcast over the gloo rendezvous of torch.distributed.run; a file; MPI ...)
 *   mcq_comm_init        collective over all ranks: ncclCommInitRank on the handle's device
 *   mcq_comm_allgather   recv [world][count] <- send [count] of every rank, device pointers, dtype MCQ_DT_*.  Asynchronous: ordered behind
 *                        everything enqueued on the handle's stream so far (the solve that filled `send`), but on a stream of its own,
 *                        so the handle's stream goes on with the NEXT solve while the gather runs (keep two send buffers and alternate).
 *                        `send` must not be overwritten and `recv` not read before mcq_comm_wait / mcq_sync.  world == 1: the same call
 *                        path (RCCL is initialised and used).
 *   mcq_comm_wait        blocks the host until the gather enqueued `lag` gathers ago has finished (0: the latest, 1: the one before:
 *                        what a caller alternating two send buffers waits for before it reuses one); ms_out (optional): its duration
 *                        on the device.  mcq_sync waits for all of them.
 *   mcq_comm_destroy     also done by mcq_destroy */
#define MCQ_COMM_ID_BYTES 128
enum { MCQ_DT_F64 = 0, MCQ_DT_F32 = 1, MCQ_DT_I32 = 2 };
int mcq_comm_unique_id(unsigned char id_out[MCQ_COMM_ID_BYTES]);
int mcq_comm_init(mcq_handle* h, int rank, int world, const unsigned char id[MCQ_COMM_ID_BYTES]);
int mcq_comm_allgather(mcq_handle* h, const void* send, void* recv, size_t count, int dtype);
int mcq_comm_wait(mcq_handle* h, int lag, float* ms_out);
int mcq_comm_world(mcq_handle* h, int* rank_out, int* world_out);      /* MCQ_E_ARG if no communicator was initialised */
int mcq_comm_destroy(mcq_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* MCQ_H */
