#!/usr/bin/env python
"""
bench.py -- headline benchmark of the MI355X minimum-curvature QP engine (contract: see the task statement).

Metric (BASELINE.json): min-curv QP solves/sec at N = 2000 waypoints, batch = 1024 per GPU.
A "step" = one pass of the hot path (assembly a1 + QP a2 + curvature-error post-check a3) over one
batch of 1024 synthetic perturbed-oval reference tracks (BASELINE config 3 generator) whose inputs
are already resident in HBM; with N > 1 GPUs every rank solves its own 1024 tracks (weak scaling, no data-path
collective inside the solve) and ONE RCCL all-gather collects the alpha vectors (north_star) inside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3|4|5] [--batch B] [--n N_WAYPOINTS] [--io f64|f32]
                  [--perturb-centreline] [--no-cpu-baseline] [--no-extras] [--force-collective]

--gpus N with N > 1 and no launcher environment (WORLD_SIZE unset): bench.py re-launches itself as N ranks, one per GPU,
under torch.distributed.run on 127.0.0.1 (what the driver does itself when it passes --gpus N under its own launcher).

--config 3 (default)  the headline workload above.
--config 4            BASELINE config 4: the lap-time matrix of 16384 (track, vehicle width, ggv scale / top speed) variants
                      over the reference's four tracks, block-partitioned over the ranks (parallel.shard_bounds), one
                      all-gather of the lap times; metric "lap-time-matrix variants/sec".
--config 5            BASELINE config 5: 65536 synthetic reference tracks over 8 GPUs = 8192 per GPU, float rows / float alpha
                      in HBM (fp64 arithmetic), per-track centrelines, one all-gather of float alpha.

Besides the contract's fields the JSON line carries (rank 0, N = 1, unless --no-extras): `host_to_host` -- the wall SURVEY.md
section 8d defines the metric on (inputs in pinned host memory -> alpha in host memory, mcq_solve_host); `iqp` -- config 3 is
mincurv_iqp: the whole iqp_handler chain of the 1024 tracks (3 passes each) as one engine call; `cpu_baseline` -- CPU-A (the
dense-faithful port of the reference's path) and CPU-B (structure-exploiting scalar solver, one problem per host core).

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INFO_DTYPE = np.dtype([("ipm_iters", "<i4"), ("as_iters", "<i4"), ("n_active_box", "<i4"), ("n_active_kappa", "<i4"),
                       ("kappa_max", "<f8"), ("kkt_res", "<f8"), ("ticks", "<i8", (8,)),
                       ("refine_rounds", "<i4"), ("second_attempt", "<i4"), ("f32_factorisations", "<i4"), ("gi_iters", "<i4")])
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (public spec; v_mfma_f64_16x16x4 runs at the fp64 VALU rate)
KAPPA_BOUND, W_VEH = 0.12, 3.4


# ----------------------------------------------------------------------------------------------------------------------
# roofline model of mcq_solve_kernel (the streamed-byte model of the saddle-point core)
# ----------------------------------------------------------------------------------------------------------------------
def work_model(n, info, band_e=32):
    """HBM bytes and fp64 flops of mcq_solve_kernel for one launch, from the iteration counts the solver reports (mcq_info).

    streamed   what THIS implementation has to move through HBM per launch -- the figure `roofline.achieved` /
               `frac` are computed from.  Since the saddle-point core (round 3, csrc/mcq_kkt.inc, mcq_tri.inc), per waypoint:
                 factorisation : elimination writes D~^-1 | Lo (160 B: D~^-1 is symmetric, 15 packed entries + Lo's five) and the
                                 forward-eliminated left spike | y (208: four of the spike's five columns, the fifth is a combination of
                                 two of them) -- G = D~^-1 Up is not stored: Up_k is Lo_(k+1)', the consumers
                                 rebuild G x from the two records -- ; the spike pass reads both back and writes the alpha rows of
                                 the spikes (64: both have rank 4, columns 1..4 each); inputs: 8 per-waypoint vectors + the mask byte (65);
                                 the right-hand side of the solve that follows rides through both passes (16)   -> 881 bytes
                 solve after a factorisation (interior-point predictor, active-set round): the separators' system (LDS) and the
                                 spike correction: spikes (64) + the vector read and written (16)              -> 80 bytes
                 any other solve (corrector, refinement round): forward chain (160 + 8 + 40), backward chain (160 + 40 + 8),
                                 correction (64 + 16)                                                          -> 496 bytes
                 float records (round 4; mcq_info.f32_factorisations of the interior-point factorisations -- the first four or five --
                                 store their records as floats): D~^-1 | Lo 80, spike | y 112 (28 floats), alpha rows 32, the forward
                                 chain's y 20:  factorisation 80 + 112 written, read back, 32 written + 65 + 16 -> 497 bytes; fused
                                 solve 32 + 16 -> 48; solve with its own chains (80 + 8 + 20) + (80 + 20 + 8) + (32 + 16) -> 264
                 gradient      : E and E' through four solves with the tridiagonal spline matrix: 27 vector accesses -> 216 bytes
                                 (the 65-wide bands of E and E' -- 1040 bytes -- no longer exist)
                 vector passes : one interior-point iteration reads / writes 33 vector entries (three passes of one load phase each;
                                 pass 1 in two halves), an active-set round ~30                                 -> 264 / 240 bytes
               interior-point iteration = 1 factorisation + predictor solve + corrector solve + passes (one exact gradient is taken where the
               float records hand over to fp64 ones; since round 5 convergence is declared on the carried gradient when every step it has
               absorbed since came from fp64 records -- rounds 2-4 confirmed with one more); active-set round = 1
               factorisation + 1 solve + 2 gradients + passes; refinement round = 1 solve + 1 gradient; + 1 initial gradient + 1 for f
               and the curvature check + 1 for the post-check.
                 assembly      : since round 4 the prologue of the same kernel (assemble_problem): rows, normals, scalings in (56 B), 13
                                 vectors out (104), the two right-hand sides written and read (32), two tridiagonal solves'
                                 factor reads (32), the derived quantities' re-reads (~32)                       -> 256 bytes
    banded     SURVEY.md section 8(d)'s figure was written for the banded-exact algorithm of rounds 1-3 (every factorisation streams the
               band of H and of L: 274 doubles per row, every sweep 144, every gradient two 65-wide bands).  The saddle-point core does
               not move those bytes; reported as `banded_model_bytes_per_launch` for reference only -- no fraction is quoted on it.

    Flops (2 per FMA), rough: elimination step ~640 FMAs per waypoint (5 x 16 working matrix, five Gauss-Jordan stages), spike pass 275,
    a chain of a solve 40 per direction, a tridiagonal solve 4.
    """
    ipm = info["ipm_iters"].astype(np.float64)
    act = info["as_iters"].astype(np.float64)
    ref = info["refine_rounds"].astype(np.float64)
    f32 = info["f32_factorisations"].astype(np.float64) if "f32_factorisations" in info.dtype.names else 0.0 * ipm
    n_fac, n_sol = ipm + act, 2 * ipm + act + ref
    # initial gradient, f + curvature check, post-check; the hand-over from float to fp64 records takes an exact gradient; the confirming one at
    # convergence is only taken when the carried gradient has absorbed float-record steps since (round 5: never, once a hand-over has happened)
    n_grad = 3.0 + 2 * act + ref + (f32 > 0) + ((f32 >= ipm) & (ipm > 0))
    ew = 2 * band_e + 1
    grad = n * 216.0
    b_fac, b_fused, b_solve = n * 881.0, n * 80.0, n * 496.0
    passes = ipm * n * 264.0 + act * n * 240.0
    plain = n_sol - n_fac                          # solves that run their own chains
    # a float-record factorisation carries one fused solve (predictor) and one solve with its own chains (corrector)
    saved = f32 * n * ((881.0 - 497.0) + (80.0 - 48.0) + (496.0 - 264.0))
    assembly = n * 256.0
    streamed = float((n_fac * (b_fac + b_fused) + plain * b_solve + n_grad * grad + passes - saved + assembly).sum())
    banded = float((n_fac * n * (130.0 + 144.0) * 8.0 + n_sol * 2.0 * n * 144.0 * 8.0 + n_grad * 2.0 * n * ew * 8.0).sum())
    flops = float((n_fac * 2.0 * n * (640.0 + 275.0) + plain * 2.0 * 2.0 * n * 40.0 + n_grad * 2.0 * 16.0 * n).sum())
    return dict(streamed=streamed, declared=banded, flops=flops,
                per_problem=dict(factorisations=float(n_fac.mean()), solves=float(n_sol.mean()), solves_with_own_chains=float(plain.mean()),
                                 gradients=float(n_grad.mean()), float_record_factorisations=float(np.mean(f32)), bytes_saved_by_float_records=float(np.mean(saved)),
                                 bytes_assembly=assembly, bytes_factorisation=b_fac, bytes_fused_solve=b_fused, bytes_solve=b_solve,
                                 bytes_gradient=grad, bytes_vector_passes=float(passes.mean())))


def source_sha():
    """SHA-256 over the engine's sources (csrc/*.hip, *.h, build.sh): what ties a counter summary under profiles/ to the binary a
    bench line was measured on (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "global_racetrajectory_optimization_amd", "csrc")
    for name in ("build.sh", "mcq_api.hip", "mcq_kernels.h", "mcq_kernels.hip", "mcq_kkt.inc", "mcq_tri.inc", "mcq_gi.inc"):
        with open(os.path.join(base, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def measured_traffic():
    """HBM traffic of mcq_solve_kernel per launch from the committed rocprofv3 PMC passes (profiles/latest_pmc.json, written by
    scripts/pmc_summary.py from separate --pmc runs of this same command; the counters cannot be read from inside an unprofiled
    run).  The summary carries the SHA-256 of the engine sources it was collected on (scripts/profile_round.sh writes it on the GPU
    box): a summary of OTHER sources is not quoted -- (None, reason)."""
    path = os.path.join(ROOT, "profiles", "latest_pmc.json")
    try:
        with open(path) as fh:
            doc = json.load(fh)
        k = doc["kernels"]["mcq_solve_kernel"]
        sha = doc.get("engine_source_sha256")
        if sha != source_sha():
            return None, "profiles/latest_pmc.json is from other engine sources (%s...): not quoted" % (sha[:12] if sha else "unstamped")
        return float(k["traffic_bytes"]), "profiles/latest_pmc.json (%s; engine sources %s...)" % (doc.get("source", "rocprofv3 --pmc passes"), sha[:12])
    except Exception:
        return None, None


# ----------------------------------------------------------------------------------------------------------------------
# GPU clock / power sampling during the timed region (explains box-to-box spread: DESIGN.md section 10)
# ----------------------------------------------------------------------------------------------------------------------
class SmiSampler(threading.Thread):
    """Polls the amdgpu hwmon nodes of one card (socket power, temperature) while the timed region runs.  Best effort: a node that
    is absent stays out of the record.  (The DPM level files report the sleep level on these boxes whatever the load -- the clock
    the kernel really ran at is measured inside it instead: config.solver_effective_sclk_mhz.)"""

    def __init__(self, index):
        super().__init__(daemon=True)
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        self.dev = os.path.dirname(cards[index]) if index < len(cards) else None
        self.samples = {"power_w": [], "temp_c": []}
        self._halt = threading.Event()

    @staticmethod
    def _cur_level(path):
        for line in open(path).read().splitlines():
            if line.rstrip().endswith("*"):
                return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        return None

    def run(self):
        if not self.dev:
            return
        hw = glob.glob(os.path.join(self.dev, "hwmon", "hwmon*"))
        while not self._halt.is_set():
            try:
                if hw:
                    for name, key, div in (("power1_average", "power_w", 1e6), ("power1_input", "power_w", 1e6), ("temp2_input", "temp_c", 1e3),
                                           ("temp1_input", "temp_c", 1e3)):
                        f = os.path.join(hw[0], name)
                        if os.path.exists(f):
                            self.samples[key].append(float(open(f).read()) / div)
            except Exception:
                pass
            self._halt.wait(0.02)

    def stop(self):
        self._halt.set()
        self.join(timeout=1.0)
        out = {}
        for k, v in self.samples.items():
            if v:
                out[k] = {"min": min(v), "mean": float(np.mean(v)), "max": max(v), "samples": len(v)}
        return out or None


# ----------------------------------------------------------------------------------------------------------------------
# CPU baselines (test infrastructure used as the checker / baseline only, after the timed region, never shipped)
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(ref_b, nv_b, sc_b, alpha_gpu, curv_gpu, a_sample, b_sample):
    """CPU-A: the port of the reference's CPU path (dense-faithful numpy assembly with BLAS on all cores + dense Goldfarb-Idnani
    in C, one thread) on the first `a_sample` problems of the workload.  CPU-B: the structure-exploiting scalar solver
    (oracle/banded_qp.c), one problem per host core, on the first `b_sample` problems.  Both are compared with the GPU's alpha."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs
    from oracle import banded_ref, qp_ref, tph_ref
    qp_ref.build()
    cores = os.cpu_count()
    n = ref_b.shape[1]
    out = {}
    # CPU-B first (seconds): built on THIS host with -O3 -march=native, one problem per hardware thread, every hardware thread
    kb = min(b_sample, ref_b.shape[0])
    banded_ref.solve_batch(ref_b[:1], nv_b[:1], sc_b[:1], KAPPA_BOUND, W_VEH, native=True)            # builds / loads the library
    t0 = time.perf_counter()
    a_b, c_b, st_b, it_b, used = banded_ref.solve_batch(ref_b[:kb], nv_b[:kb], sc_b[:kb], KAPPA_BOUND, W_VEH, nthreads=cores, native=True)
    t_b = time.perf_counter() - t0
    ok = st_b == 0
    # ... the same on half of them (one per physical core where the host has two hardware threads per core): the faster of the two is quoted
    t0 = time.perf_counter()
    _, _, _, _, used_h = banded_ref.solve_batch(ref_b[:kb], nv_b[:kb], sc_b[:kb], KAPPA_BOUND, W_VEH, nthreads=max(cores // 2, 1), native=True)
    t_h = time.perf_counter() - t0
    all_threads = {"value": kb / t_b, "cores": int(used)}
    half_threads = {"value": kb / t_h, "cores": int(used_h)}
    if t_h < t_b:
        t_b, used = t_h, used_h
    # ... and the portable build (no -march) on the runtime's default thread count, as rounds 1-3 quoted it
    banded_ref.solve_batch(ref_b[:1], nv_b[:1], sc_b[:1], KAPPA_BOUND, W_VEH)
    t0 = time.perf_counter()
    _, _, st_p, _, used_p = banded_ref.solve_batch(ref_b[:kb], nv_b[:kb], sc_b[:kb], KAPPA_BOUND, W_VEH)
    t_p = time.perf_counter() - t0
    out["cpu_b"] = {"value": kb / t_b, "unit": "solves/s", "cores": int(used), "kind": "port",
                    "sample": "%d of the %d N=%d problems, structure-exploiting scalar C (cyclic tridiagonal assembly, banded interior "
                              "point + active set, oracle/banded_qp.c built on this host with -O3 -march=native), one problem per thread, %d threads "
                              "(the faster of: every hardware thread, half of them), %.2f s" % (kb, ref_b.shape[0], n, used, t_b),
                    "every_hardware_thread": all_threads, "half_the_hardware_threads": half_threads,
                    "portable_build": {"value": kb / t_p, "cores": int(used_p), "what": "the same source without -march=native on OpenMP's default thread count"},
                    "failed": int(np.count_nonzero(~ok)),
                    "max_abs_alpha_diff_vs_gpu_m": float(np.max(np.abs(a_b[ok] - alpha_gpu[:kb][ok]))) if ok.any() else None,
                    "max_curv_err_diff_vs_gpu": float(np.max(np.abs(c_b[ok] - curv_gpu[:kb][ok]))) if ok.any() else None,
                    "mean_ipm_iters": float(it_b[:, 0].mean()), "mean_as_iters": float(it_b[:, 1].mean())}
    # CPU-A (about 10-30 s per problem at N = 2000)
    ka = min(a_sample, ref_b.shape[0])
    dt, worst, worst_c = 0.0, 0.0, 0.0
    for k in range(ka):
        A = cs.build_les_matrix(n, sc_b[k])
        t0 = time.perf_counter()
        a_ref, err_ref = tph_ref.opt_min_curv(ref_b[k], nv_b[k], A, KAPPA_BOUND, W_VEH)
        dt += time.perf_counter() - t0
        worst = max(worst, float(np.max(np.abs(a_ref - alpha_gpu[k]))))
        worst_c = max(worst_c, abs(err_ref - float(curv_gpu[k])))
    out["cpu_a"] = {"value": ka / dt, "unit": "solves/s", "cores": cores, "kind": "port",
                    "sample": "%d of the %d N=%d problems: dense-faithful numpy assembly (dense 4N x 4N inverse, BLAS threads = all %d "
                              "cores) + dense Goldfarb-Idnani in C (1 thread), %.1f s" % (ka, ref_b.shape[0], n, cores, dt),
                    "max_abs_alpha_diff_vs_gpu_m": worst, "max_curv_err_diff_vs_gpu": worst_c}
    return out


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE config 2: ONE QP behind the drop-in boundary, the way the untouched script calls it [REF main_globaltraj.py:264-271]
# ----------------------------------------------------------------------------------------------------------------------
def latency_batch1(eng, reps=15):
    """What a caller of `tph.opt_min_curv.opt_min_curv(reftrack=, normvectors=, A=, ...)` waits for, batch = 1: the drop-in's
    wall in the steady state and on the first call of a fresh handle, its pieces -- the N scalings read from the dense 4N x 4N matrix the
    reference passes, the engine call (pack + H2D + kernel + D2H), the kernel alone (HIP events) -- and, beside it on this host, the
    structure-exploiting CPU solver (oracle/banded_qp.c, ONE thread) on the same problem.  Berlin at the N of BASELINE config 2 (333), at the
    ini's default step (776), and the bench workload's N = 2000 oval; inputs from tests/golden."""
    from global_racetrajectory_optimization_amd import engine
    from global_racetrajectory_optimization_amd import trajectory_planning_helpers as tph
    from oracle import banded_ref
    root = os.path.dirname(os.path.abspath(__file__))
    saved = engine._DEFAULT_ENGINE
    out = {"what": "tph.opt_min_curv.opt_min_curv(reftrack, normvectors, A, 0.12, 3.4) through the drop-in package, one problem per call, median of "
                   "%d calls after a warm-up (ms); first_call_fresh_handle: the first call on a new engine handle in this process (workspace + "
                   "staging allocation; HIP and the code object are already loaded); cpu_b_1thread: oracle/banded_qp.c (-O3 -march=native) on one "
                   "host thread, same problem" % reps}
    try:
        for key, fixture in (("berlin_n333", "berlin_2018_n333"), ("berlin_n776", "berlin_2018"), ("oval_n2000", "oval_n2000")):
            g = np.load(os.path.join(root, "tests", "golden", fixture + ".npz"))
            ref, nv = np.ascontiguousarray(g["reftrack"]), np.ascontiguousarray(g["normvec"])
            n = ref.shape[0]
            A = tph.calc_splines.calc_splines(path=np.vstack((ref[:, :2], ref[0, :2])))[2]      # what prep_track hands the script
            fresh = engine.Engine(eng.device_id, lib_path=eng.lib_path)
            engine._DEFAULT_ENGINE = fresh
            t0 = time.perf_counter()
            a_first, _ = tph.opt_min_curv.opt_min_curv(reftrack=ref, normvectors=nv, A=A, kappa_bound=KAPPA_BOUND, w_veh=W_VEH)
            t_first = time.perf_counter() - t0
            t_call, t_sc, t_eng, k_ms = [], [], [], []
            for _ in range(reps):
                t0 = time.perf_counter()
                alpha, _ = tph.opt_min_curv.opt_min_curv(reftrack=ref, normvectors=nv, A=A, kappa_bound=KAPPA_BOUND, w_veh=W_VEH)
                t_call.append(time.perf_counter() - t0)
                k_ms.append(fresh.last_timing_ms()["solve"])
                t0 = time.perf_counter()
                sc = engine.les_scalings(A)
                t_sc.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                fresh.solve_batch([dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=KAPPA_BOUND, w_veh=W_VEH)])
                t_eng.append(time.perf_counter() - t0)
            fresh.close()
            rec = {"n": int(n), "opt_min_curv_ms": 1e3 * float(np.median(t_call)), "first_call_fresh_handle_ms": 1e3 * t_first,
                   "scalings_from_dense_A_ms": 1e3 * float(np.median(t_sc)), "engine_call_ms": 1e3 * float(np.median(t_eng)),
                   "kernel_ms": float(np.median(k_ms)), "A_bytes": int(A.nbytes),
                   "max_abs_alpha_diff_vs_golden_m": float(np.max(np.abs(alpha - g["alpha"]))),
                   "first_call_bitwise_equal": bool(np.array_equal(a_first, alpha))}
            rec["ratio_call_to_kernel"] = rec["opt_min_curv_ms"] / max(rec["kernel_ms"], 1e-9)
            try:
                scg = g["scaling"] if "scaling" in g.files else sc
                banded_ref.solve_batch(ref[None], nv[None], scg[None], KAPPA_BOUND, W_VEH, nthreads=1, native=True)
                tb = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    a_b, _, st_b, _, _ = banded_ref.solve_batch(ref[None], nv[None], scg[None], KAPPA_BOUND, W_VEH, nthreads=1, native=True)
                    tb.append(time.perf_counter() - t0)
                rec["cpu_b_1thread_ms"] = 1e3 * float(np.median(tb)) if st_b[0] == 0 else None
                rec["cpu_b_status"] = int(st_b[0])
                if st_b[0] == 0:
                    rec["cpu_b_max_abs_alpha_diff_m"] = float(np.max(np.abs(a_b[0] - alpha)))
            except Exception as e:      # (a side record must not cost the line)
                rec["cpu_b_error"] = str(e)[:200]
            out[key] = rec
    finally:
        engine._DEFAULT_ENGINE = saved
    return out


# ----------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """--gpus N > 1 without a launcher: run N ranks of this script under torch.distributed.run (one process per GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]      # "--": the launcher's own parser must not
    # look at our options (argparse would report --n as an ambiguous prefix of its --nnodes / --nproc-per-node / ...)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(cmd, env=env))


def config4_workload(rank, world, tracks=("berlin_2018", "modena_2019", "handling_track", "rounded_rectangle"), n_widths=64, nveh=64):
    """This rank's shard of BASELINE config 4: (track, vehicle width) QPs -- 4 reference tracks x 64 widths, block partition --
    each carrying its 64 (gg-scale, top-speed) vehicles.  (tracks / n_widths / nveh: the defaults ARE config 4; tests shrink them.)"""
    from global_racetrajectory_optimization_amd import parallel
    gold = [np.load(os.path.join(ROOT, "tests", "golden", t + ".npz")) for t in tracks]
    w_grid = np.linspace(2.0, 3.4, n_widths)
    uniq = [dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=KAPPA_BOUND, w_veh=float(w))
            for g in gold for w in w_grid]
    lo, hi = parallel.shard_bounds(len(uniq), world, rank)
    side = max(int(round(np.sqrt(nveh))), 1)
    v = np.arange(0.0, 72.1, 4.0)
    ggv0 = np.column_stack((v, np.full(v.size, 12.0), np.full(v.size, 12.0)))
    axm0 = np.column_stack((v, np.interp(v, [0.0, 20.0, 72.0], [5.3, 5.3, 1.2])))
    gg_scale = 0.3 + 0.7 * (np.arange(nveh) % side) / max(side - 1, 1)
    v_top = 100.0 / 3.6 + (150.0 / 3.6) * (np.arange(nveh) // side) / max((nveh - 1) // side, 1)
    mine = uniq[lo:hi]
    nvar = len(mine) * nveh
    track_of = np.repeat(np.arange(len(mine), dtype=np.int32), nveh)
    veh_of = np.tile(np.arange(nveh), len(mine))
    ggv = np.repeat(ggv0[None], nvar, axis=0)
    ggv[:, :, 1:] *= gg_scale[veh_of][:, None, None]
    return dict(qps=mine, lo=lo, n_total=len(uniq) * nveh, per_rank=-(-len(uniq) // world) * nveh, nvar=nvar, track_of=track_of, ggv=ggv,
                axm=np.repeat(axm0[None], nvar, axis=0), tops=v_top[veh_of], tracks={t: int(g["reftrack"].shape[0]) for t, g in zip(tracks, gold)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=(3, 4, 5), default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--io", choices=("f64", "f32"), default=None)
    ap.add_argument("--f32-layout", choices=("inc", "abs"), default="inc",
                    help="--io f32: float rows as ring increments + an fp64 origin per track (MCQ_F32_INCREMENTS, the default: "
                         "|alpha - alpha(fp64 rows)| <= 1e-4 m) or as absolute coordinates (MCQ_F32_ABSOLUTE: 2e-3 m at N = 2000)")
    ap.add_argument("--host-steps", type=int, default=20, help="steps of the pipelined host_to_host record")
    ap.add_argument("--c4-tracks", default="berlin_2018,modena_2019,handling_track,rounded_rectangle", help="config 4: tracks of the matrix")
    ap.add_argument("--c4-widths", type=int, default=64, help="config 4: vehicle widths per track")
    ap.add_argument("--c4-vehicles", type=int, default=64, help="config 4: (gg-scale, top-speed) vehicles per QP (a square number)")
    ap.add_argument("--perturb-centreline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host_to_host and iqp records (and the CPU baselines)")
    ap.add_argument("--cpu-a-sample", type=int, default=3)
    ap.add_argument("--cpu-b-sample", type=int, default=512)
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise RCCL and run the all-gather even with one rank (self-test of the N > 1 path on a 1-GPU box)")
    ap.add_argument("--emulate", default=None, metavar="LIBMCQ_EMU_SO",
                    help="TEST ONLY (tests/test_bench_launch.py): run the launch / sharding / collective logic on CPU tensors with the "
                         "SIMT-interpreted kernel library and the gloo backend; the line is marked invalid as a measurement")
    args = ap.parse_args()
    if args.config == 5:
        args.batch = args.batch or 8192
        args.io = args.io or "f32"
        args.perturb_centreline = True
    args.batch = args.batch or 1024
    args.io = args.io or "f64"
    if args.emulate and os.environ.get("MCQ_BENCH_TEST_N"):      # tests/test_bench_launch.py under a launcher whose parser rejects --n
        args.n = int(os.environ["MCQ_BENCH_TEST_N"])

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        self_launch(args)
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting the ranks that exist" % (args.gpus, world), file=sys.stderr)

    from global_racetrajectory_optimization_amd import engine, parallel, synthetic

    emulate = args.emulate is not None
    # Nothing of torch touches the GPU (round 4): device memory, streams, events and the collective are the engine's (C ABI); torch is
    # imported for torch.distributed alone -- the launcher's rendezvous (gloo), its barriers and the exchange of the ranks' wall times.
    try:
        eng = engine.Engine(0 if emulate else local_rank, lib_path=args.emulate)
    except engine.EngineError as e:
        # a launcher that hands every rank ONE visible device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per process): that device is 0
        if emulate or local_rank == 0 or "no such device" not in str(e):
            raise SystemExit("bench.py needs a GPU for rank %d (the engine has no CPU path): %s" % (rank, e))
        try:
            eng = engine.Engine(0, lib_path=args.emulate)
        except engine.EngineError as e2:
            raise SystemExit("bench.py needs a GPU for rank %d (the engine has no CPU path): %s" % (rank, e2))

    dist = None
    collective = world > 1 or args.force_collective
    if collective:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints its version banner (NCCL_DEBUG=VERSION is set on the GPU boxes) with printf when it comes up: keep it
        # off stdout, where exactly ONE JSON line is expected -- file descriptor 1 points at stderr meanwhile
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            parallel.init_engine_comm(eng, dist)              # ncclCommInitRank behind the C ABI; the id travels over gloo
            dist.barrier()
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    B, n = args.batch, args.n
    f32 = args.io == "f32"
    io_np = np.float32 if f32 else np.float64
    io_dt = eng.DT_F32 if f32 else eng.DT_F64
    solve_ms = []
    ag_ms_list = []     # device time of the gathers of the timed region (mcq_comm_wait)

    def dbuf(arr=None, nbytes=0):
        p = eng.alloc(arr.nbytes if arr is not None else nbytes)
        if arr is not None:
            eng.upload(p, arr)
        return p

    def gather(d_send, d_recv, count, np_dt, dt_code, record):
        """The one collective of a step: RCCL through the C ABI (asynchronous, the engine's comm stream).  (The CPU tests run this very
        path: the SIMT-interpreted library loads tests/stub/librccl_stub_sync.so through $MCQ_RCCL_LIB.)"""
        eng.comm_allgather(d_send, d_recv, count, dt_code)

    if args.config == 4:
        wl = config4_workload(rank, world, tuple(args.c4_tracks.split(",")), args.c4_widths, args.c4_vehicles)
        d_lap = dbuf(nbytes=8 * wl["per_rank"])
        d_all = dbuf(nbytes=8 * world * wl["per_rank"]) if collective else None
        lap_h = [None]

        def step(record):
            al, _, st, _ = eng.solve_batch(wl["qps"])
            if np.any(st != 0):
                raise SystemExit("config 4: QP status %s" % np.unique(st))
            race = eng.raceline_batch([p["reftrack"] for p in wl["qps"]], [p["normvec"] for p in wl["qps"]], al, 2.0)
            _, lap = eng.vel_profile_batch(race["kappa"], race["el_lengths"], wl["ggv"], wl["axm"], 0.75, 1200.0, wl["tops"], 1.0,
                                           track_of=wl["track_of"], n_of_track=race["m"])
            lap_h[0] = lap
            if collective:
                ms = eng.comm_wait(0)              # the previous gather reads d_lap: done before it is overwritten
                if record and ms > 0.0:
                    ag_ms_list.append(ms)
                pad = np.zeros(wl["per_rank"])
                pad[:lap.size] = lap
                eng.upload(d_lap, pad)
                gather(d_lap, d_all, wl["per_rank"], np.float64, eng.DT_F64, record)
    else:
        ref_h, nv_h, sc_h = synthetic.oval_batch(B, n=n, first=rank * B, perturb_centreline=args.perturb_centreline)
        d_org = None
        if f32 and args.f32_layout == "inc":
            rows32, org = engine.rows_to_increments(ref_h)           # float ring increments + fp64 origin per track
            d_ref = dbuf(rows32)
            d_org = dbuf(org)
        else:
            d_ref = dbuf(ref_h.astype(io_np))
        d_nv = dbuf(nv_h.astype(io_np))
        d_sc = dbuf(sc_h.astype(io_np))
        # two result buffers, used in turn: the all-gather of step k (the engine's comm stream) runs while step k+1 solves into the other one
        item = np.dtype(io_np).itemsize
        d_alpha2 = [dbuf(nbytes=item * B * n) for _ in range(2 if collective else 1)]
        d_alpha = d_alpha2[0]
        step_no = [0]
        d_curv = dbuf(nbytes=8 * B)
        d_status = dbuf(nbytes=4 * B)
        d_info = dbuf(nbytes=INFO_DTYPE.itemsize * B)
        d_all = dbuf(nbytes=item * world * B * n) if collective else None

        def step(record):
            slot = step_no[0] % len(d_alpha2)
            d_alpha = d_alpha2[slot]
            step_no[0] += 1
            if collective:
                ms = eng.comm_wait(1)              # the gather that last read this buffer (two steps ago) has finished
                if record and ms > 0.0:
                    ag_ms_list.append(ms)
            if f32:     # float normals are unit vectors only to 6e-8: the engine derives them (and the scalings) in fp64
                eng.solve_device_f32_rows(B, n, engine.F32_INCREMENTS if d_org is not None else engine.F32_ABSOLUTE, d_ref,
                                          d_org, KAPPA_BOUND, W_VEH, d_alpha, d_curv, d_status, d_info)
            else:
                eng.solve_device(B, n, d_ref, d_nv, d_sc, KAPPA_BOUND, W_VEH, d_alpha, d_curv, d_status, d_info)
            if collective:
                gather(d_alpha, d_all, B * n, io_np, io_dt, record)

    def fence():
        eng.sync()                  # the engine's stream AND its comm stream (the HIP device of this rank is idle afterwards)
        if collective:
            dist.barrier()

    for _ in range(args.warmup):
        step(False)
    fence()
    sampler = SmiSampler(local_rank) if (rank == 0 and not emulate) else None
    if sampler:
        sampler.start()
    # the K steps are enqueued back to back -- no host synchronisation inside the timed region (until round 5 every step synchronised to read its
    # kernel's HIP-event time, and a slow host core showed up in `value`); the launches of the region are timed ON THE DEVICE as one span
    # (mcq_timing_begin / mcq_timing_end: events on the engine's compute stream): span / launches = the average duration of a launch
    if args.config != 4:
        eng.timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    if args.config != 4:
        span_ms, span_launches = eng.timing_end()          # (waits for the last launch of the region)
        solve_ms.append({"solve": span_ms / max(span_launches, 1), "total": span_ms / max(span_launches, 1), "last_launch": eng.last_timing_ms()["solve"],
                         "launches": span_launches})
    fence()
    dt_local = time.perf_counter() - t0
    if args.config != 4:
        d_alpha = d_alpha2[(step_no[0] - 1) % len(d_alpha2)]     # the buffer of the last step
    clocks = sampler.stop() if sampler else None
    dt, rank_ms = dt_local, [1e3 * dt_local / args.steps]
    if collective:
        t = torch.tensor([dt_local], dtype=torch.float64)
        allt = torch.zeros((world,), dtype=torch.float64)
        dist.all_gather_into_tensor(allt, t)
        rank_ms = [1e3 * float(v) / args.steps for v in allt]
        dt = float(allt.max().item())
        if args.config != 4:
            own = eng.download(d_all, (B, n), io_np, offset_bytes=rank * B * n * item)
            assert np.array_equal(own, eng.download(d_alpha, (B, n), io_np)), "all-gather: own shard differs"
    ranks_seen = dist.get_world_size() if collective else 1
    assert ranks_seen == world, "process group has %d ranks, the launcher announced %d" % (ranks_seen, world)
    if collective:
        assert eng.comm_world() == (rank, world), "the engine's RCCL communicator is %s, the launcher announced rank %d of %d" % (eng.comm_world(), rank, world)
    ag_ms = float(np.mean(ag_ms_list)) if ag_ms_list else None

    out = None
    if args.config == 4:
        # every rank's lap times in partition order (the gathered tensor is padded to the largest shard); listed in the line for
        # small matrices only (tests)
        gathered_laps = None
        if wl["n_total"] <= 256:
            if collective:
                allv = eng.download(d_all, (world, wl["per_rank"]), np.float64)
                _par = parallel
                nq = wl["n_total"] // args.c4_vehicles
                gathered_laps = [float(v) for r in range(world)
                                 for v in allv[r, :(_par.shard_bounds(nq, world, r)[1] - _par.shard_bounds(nq, world, r)[0]) * args.c4_vehicles]]
            else:
                gathered_laps = [float(v) for v in lap_h[0]]
        if rank == 0:
            lap = lap_h[0]
            out = {"metric": "lap-time-matrix variants/sec (BASELINE config 4: 16384 variants)", "value": wl["n_total"] * args.steps / dt,
                   "unit": "variants/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                   "data": "reference tracks (tests/golden), synthetic vehicles" + (" -- EMULATED ON CPU (test of the launch logic): NOT A MEASUREMENT" if emulate else ""),
                   "config": {"workload": "BASELINE config 4: lap-time matrix of %d variants = 4 reference tracks x 64 vehicle widths (256 QPs, "
                                          "solved once each: the vehicle tables do not enter the QP) x 64 (gg-scale, top-speed) vehicles; per step: "
                                          "QPs + racelines + velocity profiles through the host-buffer entries (packing + PCIe included), block "
                                          "partition over the ranks, one all-gather of the lap times" % wl["n_total"],
                              "tracks": wl["tracks"], "variants_total": wl["n_total"], "variants_this_rank": wl["nvar"], "ranks_seen": ranks_seen,
                              "lap_times_gathered_s": gathered_laps,
                              "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "by_rank": rank_ms},
                              "allgather_ms": ag_ms,
                              "lap_time_range_s": [float(lap.min()), float(lap.max())]}}
    elif rank == 0:
        status = eng.download(d_status, (B,), np.int32)
        info = eng.download(d_info, (B,), INFO_DTYPE)
        alpha_gpu = eng.download(d_alpha, (B, n), io_np).astype(np.float64)
        curv_gpu = eng.download(d_curv, (B,), np.float64)
        value = world * B * args.steps / dt
        k_ms = float(np.mean([m["solve"] for m in solve_ms]))
        wm = work_model(n, info)
        achieved = wm["streamed"] / (k_ms * 1e-3) / 1e9
        default_wl = B == 1024 and n == 2000 and not args.perturb_centreline
        traffic, traffic_src = measured_traffic() if default_wl else (None, None)
        out = {
            "metric": "min-curv QP solves/sec, N=2000 waypoints, batch=1024 per GPU",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not emulate else "synthetic -- EMULATED ON CPU (test of the launch logic): NOT A MEASUREMENT",
            "config": {"workload": "BASELINE config %s generator: perturbed 2:1 oval, perimeter 6000 m, N=%d, batch=%d "
                                   "%s per GPU, one opt_min_curv pass (assembly + QP + "
                                   "curvature-error check) per track per step; kappa_bound=0.12, w_veh=3.4"
                                   % ("5" if args.perturb_centreline else "3", n, B,
                                      "reference tracks (centreline and widths perturbed per track)" if args.perturb_centreline
                                      else "track-width perturbations"),
                       "batch_per_gpu": B, "n_waypoints": n,
                       "io": args.io + ((" rows (%s) / float alpha in HBM, fp64 arithmetic" % ("ring increments + fp64 origin" if args.f32_layout == "inc" else "absolute coordinates")) if f32 else ""),
                       "centrelines": "perturbed per track" if args.perturb_centreline else "shared",
                       "collective": ("1 all-gather of alpha per step: ncclAllGather (RCCL) through the C ABI, mcq_comm_allgather, on the engine's comm stream"
                                      + (" [EMULATED RUN: librccl stand-in " + os.path.basename(os.environ.get("MCQ_RCCL_LIB", "?")) + "]" if emulate else "")
                                      if collective else "none"),
                       "ranks_seen": ranks_seen, "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms), "by_rank": rank_ms},
                       "allgather_ms": ag_ms, "allgather_dtype": np.dtype(io_np).name,
                       "failed_problems": int(np.count_nonzero(status)),
                       "mean_ipm_iters": float(info["ipm_iters"].mean()), "mean_as_iters": float(info["as_iters"].mean()),
                       "mean_active_box_rows": float(info["n_active_box"].mean()),
                       "mean_refine_rounds": float(info["refine_rounds"].mean()),
                       "second_attempts": int(np.count_nonzero(info["second_attempt"] & 1)), "warm_start_fallbacks": int(np.count_nonzero(info["second_attempt"] & 2)),
                       "kernel_ms": {"solve": k_ms, "what": "average duration of a launch over the timed region, on the device (mcq_timing_begin / _end: %d launches "
                                                                 "back to back on the engine's stream)" % solve_ms[0]["launches"],
                                     "last_launch_alone": solve_ms[0]["last_launch"]},
                       "goldfarb_idnani_fallbacks": int(np.count_nonzero(info["gi_iters"] > 0)),
                       "workspace_GB": eng.workspace_bytes() / 1e9,
                       "solver_phase_ms_per_problem": {k: float(info["ticks"][:, j].mean()) / 1e5 for j, k in
                                                       enumerate(("factor", "solve", "gradient", "kernel", "interior_point_phase", "active_set_phase"))},
                       "ticks_mean": [float(v) for v in info["ticks"].mean(axis=0)],
                       # effective shader clock of the solver kernel: s_memtime / s_memrealtime read inside the kernel, per problem
                       "solver_effective_sclk_mhz": {"mean": float(np.mean(100.0 * info["ticks"][:, 6] / np.maximum(info["ticks"][:, 3], 1))),
                                                     "min": float(np.min(100.0 * info["ticks"][:, 6] / np.maximum(info["ticks"][:, 3], 1)))},
                       "engine_source_sha256": source_sha(),
                       "gpu_power_temp_during_timed_region": clocks},
            # `achieved` / `frac`: the bytes this implementation has to stream (work_model: the records of the saddle-point elimination as
            # laid out in HBM) over the kernel's average duration (HIP events on the engine's stream, this run).
            "roofline": {"bound": "hbm", "kernel": "mcq_solve_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": wm["streamed"], "kernel_ms": k_ms,
                         "model": "streamed (DESIGN.md section 6); per problem: %s" % json.dumps({k: round(v, 3) for k, v in wm["per_problem"].items()}),
                         "frac_of_measured_traffic": (traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         # the bytes the banded-exact algorithm of rounds 1-3 would have moved for the same iteration
                         # counts: for reference, no fraction quoted -- the saddle-point core does not move them
                         "banded_model_bytes_per_launch": wm["declared"],
                         "durations": "achieved / frac: kernel_ms of THIS run (device-side span over the timed region's launches); traffic: counters of "
                                      "the profiled run named in traffic_source (its own kernel duration is in that summary; frac_of_measured_traffic "
                                      "divides its bytes by THIS run's kernel_ms)",
                         "mfma": "SQ_INSTS_MFMA = 0 by design (profiles/*_pmc.md): H = E'E is never formed since round 4 -- every linear system is the 5 x 5-block "
                                 "saddle-point elimination or a scalar tridiagonal sweep, far below a 16 x 16 x 4 MFMA tile; north_star's 'MFMA for the dense "
                                 "H = M'M contraction' has no contraction left to run on (DESIGN.md section 1); the bound that applies is HBM",
                         "fp64_flops_per_launch": wm["flops"], "fp64_tflops": wm["flops"] / (k_ms * 1e-3) / 1e12,
                         "fp64_frac_of_%.1f_tflops" % FP64_PEAK_TFLOPS: wm["flops"] / (k_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
        }
        extras = world == 1 and not args.no_extras and not f32 and not emulate
        if extras:
            # ---- the wall SURVEY.md section 8d defines: inputs in pinned host memory -> alpha in host memory ---------------------
            # blocking entry (one batch: H2D -> kernels -> D2H) and the pipelined stream of batches (mcq_solve_host_pipelined: step k's
            # kernels run while step k+1 uploads and step k-1 downloads)
            p_ref, p_nv, p_sc = eng.host_array((B, n, 4)), eng.host_array((B, n, 2)), eng.host_array((B, n))
            p_al = [eng.host_array((B, n)), eng.host_array((B, n))]
            p_ref[...], p_nv[...], p_sc[...] = ref_h, nv_h, sc_h
            eng.solve_host(p_ref, p_nv, p_sc, KAPPA_BOUND, W_VEH, alpha_out=p_al[0])
            hh = []
            for _ in range(3):
                t1 = time.perf_counter()
                _, _, st_h, _ = eng.solve_host(p_ref, p_nv, p_sc, KAPPA_BOUND, W_VEH, alpha_out=p_al[0])
                hh.append(time.perf_counter() - t1)
            os.environ["MCQ_HOST_ONE_LAUNCH"] = "1"          # (A/B: the same entry as ONE upload -> launch -> download, as rounds 1-5 ran it)
            hh1 = []
            for _ in range(3):
                t1 = time.perf_counter()
                eng.solve_host(p_ref, p_nv, p_sc, KAPPA_BOUND, W_VEH, alpha_out=p_al[1])
                hh1.append(time.perf_counter() - t1)
            del os.environ["MCQ_HOST_ONE_LAUNCH"]
            blocking_equal = bool(np.array_equal(p_al[0], p_al[1]))
            hs = max(args.host_steps, 2)
            eng.solve_host_pipelined([p_ref] * 2, [p_nv] * 2, [p_sc] * 2, KAPPA_BOUND, W_VEH, p_al)       # staging slots, streams
            p_al[0][...] = np.nan
            p_al[1][...] = np.nan
            t1 = time.perf_counter()
            _, st_p = eng.solve_host_pipelined([p_ref] * hs, [p_nv] * hs, [p_sc] * hs, KAPPA_BOUND, W_VEH, [p_al[k & 1] for k in range(hs)])
            t_pipe = time.perf_counter() - t1
            out["host_to_host"] = {"value": B * hs / t_pipe, "unit": "solves/s", "ms_per_step": 1e3 * t_pipe / hs, "steps": hs,
                                   "what": "mcq_solve_host_pipelined: a stream of %d batches, rows / normals / scalings in pinned host memory -> alpha, "
                                           "curv_error, status in pinned host memory; uploads, kernels and downloads of consecutive batches overlap "
                                           "(two copy streams, two staging slots) and -- round 5 -- so do the KERNELS of consecutive batches (two compute "
                                           "streams, two workspaces: the next launch starts on the compute units the previous one has left), which is why "
                                           "this rate can exceed `value`, measured launch by launch on one stream" % hs,
                                   "ratio_to_device_resident": (B * hs / t_pipe) / value,
                                   "blocking_single_batch": {"value": B / float(np.mean(hh)), "ms_per_step": 1e3 * float(np.mean(hh)),
                                                             "what": "mcq_solve_host, ONE batch, blocking; round 6: in two slices, one per compute stream -- upload k + 1 / kernel k / "
                                                                     "download k - 1 overlapped; %d calls" % len(hh),
                                                             "as_one_launch": {"value": B / float(np.mean(hh1)), "ms_per_step": 1e3 * float(np.mean(hh1)),
                                                                               "what": "$MCQ_HOST_ONE_LAUNCH=1: H2D -> one launch -> D2H (rounds 1-5)"},
                                                             "alpha_equal_between_the_two": blocking_equal},
                                   "bytes_h2d_per_step": int(p_ref.nbytes + p_nv.nbytes + p_sc.nbytes), "bytes_d2h_per_step": int(p_al[0].nbytes),
                                   "alpha_equal_to_device_resident_run": bool(np.array_equal(p_al[0], alpha_gpu) and np.array_equal(p_al[1], alpha_gpu)),
                                   "failed_problems": int(np.count_nonzero(st_h) + np.count_nonzero(st_p))}
            # ---- the tail of a launch: the same device-resident step with the launches of CONSECUTIVE steps on two streams (two handles, a
            #      workspace each): step k + 1 starts on the compute units step k has left while its slowest problems finish -- what
            #      mcq_solve_host_pipelined does for batches from host memory.  A side record: `value` stays the launch-by-launch rate.
            try:
                hs2 = 2 * max(args.steps, 5)
                d_al2 = [d_alpha, eng.alloc(8 * B * n)]
                d_cu2 = [d_curv, eng.alloc(8 * B)]
                d_st2 = [d_status, eng.alloc(4 * B)]
                lists = lambda k_: ([d_ref] * k_, [d_nv] * k_, [d_sc] * k_, [d_al2[q & 1] for q in range(k_)], [d_cu2[q & 1] for q in range(k_)],
                                    [d_st2[q & 1] for q in range(k_)])
                r_, v_, s_, a_, c_, t_ = lists(4)
                eng.solve_device_stream(B, n, r_, v_, s_, KAPPA_BOUND, W_VEH, a_, c_, t_)
                eng.sync()
                r_, v_, s_, a_, c_, t_ = lists(hs2)
                t2 = time.perf_counter()
                eng.solve_device_stream(B, n, r_, v_, s_, KAPPA_BOUND, W_VEH, a_, c_, t_)
                eng.sync()
                t_two = (time.perf_counter() - t2) / hs2
                out["device_resident_two_streams"] = {
                    "value": B / t_two, "unit": "solves/s", "ms_per_step": 1e3 * t_two, "steps": hs2, "ratio_to_value": (B / t_two) / value,
                    "alpha_equal": bool(np.array_equal(eng.download(d_al2[1], (B, n), np.float64), alpha_gpu)
                                        and np.array_equal(eng.download(d_al2[0], (B, n), np.float64), alpha_gpu)),
                    "what": "mcq_solve_device_stream: the timed loop of `value` as ONE call that alternates consecutive steps between the engine's two "
                            "compute streams (a workspace each): the launches overlap, so a launch's tail -- 10.4 ms against ~9.4 ms of mean load -- is "
                            "filled by the next one's workgroups.  Not the headline: with two launches in flight a kernel's own duration no longer "
                            "measures a step"}
            except Exception as e:      # (a side record must not cost the line)
                out["device_resident_two_streams"] = {"error": str(e)[:200]}
            # ---- config 3 is mincurv_iqp: the whole iqp_handler chain of the same tracks as one engine call ------------------------
            trk = dict(reftrack=ref_h, normvectors=nv_h, scaling=sc_h)      # (stacked arrays: row k is track k)
            w0 = eng.iqp_batch(trk, KAPPA_BOUND, W_VEH, 3.0)  # first call: workspace + pinned staging of this size are allocated
            nmx = w0["stats"]["nmax"]
            # end states into page-locked arrays the caller keeps across calls (two calls timed: the fresh-array path is the one a
            # one-off caller pays -- its arrays are touched inside the call -- the second is the steady state of a batch service)
            obuf = dict(alpha=eng.host_array((B, nmx)), reftrack=eng.host_array((B, nmx, 4)), normvectors=eng.host_array((B, nmx, 2)))
            t1 = time.perf_counter()
            iq0 = eng.iqp_batch(trk, KAPPA_BOUND, W_VEH, 3.0, iters_min=3, curv_error_allowed=0.01, nmax=nmx)
            t_iqp_fresh = time.perf_counter() - t1
            t1 = time.perf_counter()
            iq_pg = eng.iqp_batch(trk, KAPPA_BOUND, W_VEH, 3.0, iters_min=3, curv_error_allowed=0.01, nmax=nmx, out=obuf)
            t_iqp_pageable = time.perf_counter() - t1
            # the steady state of a batch service: its input arrays are page-locked too (the rows of p_ref / p_nv / p_sc, back to back) -- round 6: such
            # a batch goes to the device without the packing pass (mcq_last_upload_was_direct)
            trk_p = dict(reftrack=p_ref, normvectors=p_nv, scaling=p_sc)
            eng.iqp_batch(trk_p, KAPPA_BOUND, W_VEH, 3.0, iters_min=3, curv_error_allowed=0.01, nmax=nmx, out=obuf)
            t1 = time.perf_counter()
            iq = eng.iqp_batch(trk_p, KAPPA_BOUND, W_VEH, 3.0, iters_min=3, curv_error_allowed=0.01, nmax=nmx, out=obuf)
            t_iqp = time.perf_counter() - t1
            direct = eng.last_upload_was_direct()
            same = all(np.array_equal(a, b) for a, b in zip(iq0["alpha"], iq["alpha"])) and all(np.array_equal(a, b) for a, b in zip(iq0["alpha"], iq_pg["alpha"]))
            # per-pass figures need the host between the rounds: the same call with timed statistics (one launch per round over the whole
            # batch, per-track records read back after every pass) -- its end states must be the same bit for bit
            t1 = time.perf_counter()
            iqt = eng.iqp_batch(trk, KAPPA_BOUND, W_VEH, 3.0, iters_min=3, curv_error_allowed=0.01, timed=True, nmax=nmx)
            t_iqp_timed = time.perf_counter() - t1
            same = same and all(np.array_equal(a, b) for a, b in zip(iq0["alpha"], iqt["alpha"]))
            out["iqp"] = {"value": iq["stats"]["qp_solves"] / t_iqp, "unit": "QP solves/s (end to end: host tracks in, end states out)",
                          "tracks": B, "seconds": t_iqp, "rounds": iq["stats"]["rounds"], "qp_solves": iq["stats"]["qp_solves"],
                          "pass_ms": iqt["stats"]["solver_ms"], "warm_start_fallbacks_per_pass": iqt["stats"]["fallbacks"],
                          "seconds_round_by_round_with_statistics": t_iqp_timed,
                          "failed_tracks": int(np.count_nonzero(iq["status"])), "n_final_range": [int(iq["n"].min()), int(iq["n"].max())],
                          "seconds_fresh_output_arrays": t_iqp_fresh, "seconds_pageable_inputs": t_iqp_pageable, "inputs_uploaded_without_packing": bool(direct),
                          "alpha_equal_between_the_calls": bool(same),
                          "what": "mcq_iqp_batch: iqp_handler (stepsize_interp 3.0, iters_min 3, curv_error_allowed 0.01) of the %d tracks as one "
                                  "call -- QP passes, termination test, damping and re-linearisation glue on the device, passes 2+ warm-started; "
                                  "the first iters_min rounds are ONE launch in which every workgroup takes its track through the rounds on its "
                                  "own (mcq_iqp_rounds_kernel); inputs AND end states in page-locked arrays kept by the caller: the batch goes up straight "
                                  "from them, no packing pass (round 6; seconds_pageable_inputs: the same call from pageable numpy inputs, packed into "
                                  "pinned staging by several host threads; seconds_fresh_output_arrays: also allocating and touching fresh output "
                                  "arrays).  pass_ms / fallbacks: from a third call with per-round statistics (one launch per round, HIP "
                                  "events, records read back after every pass: seconds_round_by_round_with_statistics)" % B}
            if not args.no_cpu_baseline:
                cb = cpu_baseline(ref_h, nv_h, sc_h, alpha_gpu, curv_gpu, args.cpu_a_sample, args.cpu_b_sample)
                out["cpu_baseline"] = dict(cb["cpu_a"], also={"cpu_b": cb["cpu_b"]})
                try:
                    out["latency_batch1"] = latency_batch1(eng)
                except Exception as e:      # (a side record must not cost the line)
                    out["latency_batch1"] = {"error": str(e)[:300]}
        elif world == 1 and not args.no_cpu_baseline and not args.no_extras and not emulate:
            # f32 boundary: the baseline solves the rows the engine saw
            r32 = engine.increments_to_rows(rows32, org) if d_org is not None else ref_h.astype(np.float32).astype(np.float64)
            nv32 = np.stack([synthetic.prepared_track(r32[k, :, :2])[0] for k in range(min(B, args.cpu_b_sample))])
            sc32 = np.stack([synthetic.prepared_track(r32[k, :, :2])[1] for k in range(min(B, args.cpu_b_sample))])
            cb = cpu_baseline(r32[:nv32.shape[0]], nv32, sc32, alpha_gpu, curv_gpu, args.cpu_a_sample, args.cpu_b_sample)
            out["cpu_baseline"] = dict(cb["cpu_a"], also={"cpu_b": cb["cpu_b"]})
    if out is not None:
        print(json.dumps(out))
        sys.stdout.flush()
    if collective:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
