#!/usr/bin/env python
"""
bench.py -- headline benchmark of the MI355X minimum-curvature QP engine (contract: see the task statement).

Metric (BASELINE.json): min-curv QP solves/sec at N = 2000 waypoints, batch = 1024 per GPU.
A "step" = one pass of the hot path (assembly a1 + QP a2 + curvature-error post-check a3) over one
batch of 1024 synthetic perturbed-oval reference tracks (BASELINE config 3 generator) whose inputs
are already resident in HBM; with N > 1 GPUs every rank solves its own 1024 tracks (weak scaling, no data-path
collective inside the solve) and ONE RCCL all-gather collects the alpha vectors (north_star) inside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--n N_WAYPOINTS] [--io f64|f32] [--perturb-centreline]
                  [--no-cpu-baseline]

--io f32 (BASELINE config 5's boundary): tracks and alpha live in HBM as float and the all-gather moves float alpha; the
arithmetic stays fp64 (`dtype` in the JSON line is the arithmetic type).  Config 5 on one node of 8 GPUs is
`--gpus 8 --batch 8192 --io f32 --perturb-centreline` (65536 synthetic reference tracks, 69 GB of workspace per GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INFO_DTYPE = np.dtype([("ipm_iters", "<i4"), ("as_iters", "<i4"), ("n_active_box", "<i4"), ("n_active_kappa", "<i4"),
                       ("kappa_max", "<f8"), ("kkt_res", "<f8"), ("ticks", "<i8", (8,)),
                       ("refine_rounds", "<i4"), ("second_attempt", "<i4")])
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KAPPA_BOUND, W_VEH = 0.12, 3.4


def algorithmic_bytes(n, info, band_e=32):
    """Algorithmic HBM bytes of mcq_solve_kernel for one launch ('banded-exact' mode).

    Row sizes as stored (csrc/mcq_kernels.h): H row 130 doubles (65 band | pad | 64 border), L row 144 doubles
    (64 band | 16 inverse-diagonal-tile | 64 border W), E / E' bands 65 doubles per row.
      factorisation : read the H rows + write the L rows               n * (1040 + 1152) B
      solve         : forward + backward sweep, each reads the L rows  2 * n * 1152 B
      gradient      : E band + E' band                                 2 * n * 65 * 8 B
    IPM iteration = 1 factorisation + 2 solves (the gradient is carried through the reduced system; one exact gradient
    confirms convergence); active-set iteration = 1 factorisation + 1 solve + 2 gradients; refinement round = 1 solve +
    1 gradient (the rounds actually run: mcq_info.refine_rounds); + 1 initial gradient + 3 band products in the epilogue.
    Iteration counts are the ones the solver reports (mcq_info).
    """
    fac = n * (130.0 + 144.0) * 8.0
    sol = 2.0 * n * 144.0 * 8.0
    grad = 2.0 * n * (2 * band_e + 1) * 8.0
    ipm = info["ipm_iters"].astype(np.float64)
    act = info["as_iters"].astype(np.float64)
    ref = info["refine_rounds"].astype(np.float64)
    per = ipm * (fac + 2 * sol) + grad + act * (fac + sol + 2 * grad) + ref * (sol + grad) + grad + 1.5 * grad
    return float(per.sum())


def measured_traffic():
    """HBM traffic of mcq_solve_kernel per launch from the committed rocprofv3 PMC passes (profiles/latest_pmc.json,
    written by scripts/pmc_summary.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, KiB units,
    FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  None if no profile is committed."""
    path = os.path.join(ROOT, "profiles", "latest_pmc.json")
    try:
        with open(path) as fh:
            k = json.load(fh)["kernels"]["mcq_solve_kernel"]
        return float(k["traffic_bytes"])
    except Exception:
        return None


CPU_SAMPLE = 2       # problems of the workload the CPU baseline is timed on (~10 s each at N = 2000)


def cpu_baseline(ref_b, nv_b, sc_b):
    """Oracle ('port' of the reference's CPU path: dense-faithful numpy assembly + dense Goldfarb-Idnani in C) timed on
    the first CPU_SAMPLE problems of the same workload.  Test infrastructure used as the checker/baseline only, never
    shipped.  Returns (alpha of problem 0, curvature error of problem 0, total seconds, problems timed)."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs
    from oracle import qp_ref, tph_ref
    qp_ref.build()
    k_max = min(CPU_SAMPLE, ref_b.shape[0])
    out0, dt = None, 0.0
    for k in range(k_max):
        A = cs.build_les_matrix(ref_b[k].shape[0], sc_b[k])
        t0 = time.perf_counter()
        res = tph_ref.opt_min_curv(ref_b[k], nv_b[k], A, KAPPA_BOUND, W_VEH)
        dt += time.perf_counter() - t0
        if k == 0:
            out0 = res
    return out0[0], out0[1], dt, k_max


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--io", choices=("f64", "f32"), default="f64")
    ap.add_argument("--perturb-centreline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise RCCL and run the all-gather even with one rank (self-test of the N > 1 path on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    from global_racetrajectory_optimization_amd import engine, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    collective = world > 1 or args.force_collective
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints its version banner (NCCL_DEBUG=VERSION is set on the GPU boxes) with printf on the first collective: keep it
        # off stdout, where exactly ONE JSON line is expected -- file descriptor 1 points at stderr while RCCL comes up
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    B, n = args.batch, args.n
    ref_h, nv_h, sc_h = synthetic.oval_batch(B, n=n, first=rank * B, perturb_centreline=args.perturb_centreline)
    f32 = args.io == "f32"
    io_t = torch.float32 if f32 else torch.float64
    d_ref = torch.from_numpy(ref_h).to(dev).to(io_t)
    d_nv = torch.from_numpy(nv_h).to(dev).to(io_t)
    d_sc = torch.from_numpy(sc_h).to(dev).to(io_t)
    d_alpha = torch.zeros((B, n), dtype=io_t, device=dev)
    d_curv = torch.zeros((B,), dtype=torch.float64, device=dev)
    d_status = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_info = torch.zeros((B, INFO_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    d_all = torch.zeros((world * B, n), dtype=io_t, device=dev) if collective else None

    eng = engine.Engine(local_rank)
    solve_ms = []

    def step(record):
        if f32:     # float normals are unit vectors only to 6e-8: let the engine derive them (and the scalings) in fp64
            eng.solve_device_f32(B, n, d_ref.data_ptr(), None, None, KAPPA_BOUND, W_VEH,
                                 d_alpha.data_ptr(), d_curv.data_ptr(), d_status.data_ptr(), d_info.data_ptr())
        else:
            eng.solve_device(B, n, d_ref.data_ptr(), d_nv.data_ptr(), d_sc.data_ptr(), KAPPA_BOUND, W_VEH,
                             d_alpha.data_ptr(), d_curv.data_ptr(), d_status.data_ptr(), d_info.data_ptr())
        eng.sync()
        if record:
            solve_ms.append(eng.last_timing_ms())
        if collective:
            dist.all_gather_into_tensor(d_all, d_alpha)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if collective:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        assert torch.equal(d_all[rank * B:(rank + 1) * B], d_alpha), "all-gather: own shard differs"

    status = d_status.cpu().numpy()
    info = d_info.cpu().numpy().view(INFO_DTYPE).reshape(B)
    n_bad = int(np.count_nonzero(status))
    alpha0 = d_alpha[0].cpu().numpy().astype(np.float64)
    curv0 = float(d_curv[0].item())

    if rank == 0:
        value = world * B * args.steps / dt
        k_ms = float(np.mean([m["solve"] for m in solve_ms]))
        alg = algorithmic_bytes(n, info)
        achieved = alg / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "min-curv QP solves/sec, N=2000 waypoints, batch=1024 per GPU",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config %s generator: perturbed 2:1 oval, perimeter 6000 m, N=%d, batch=%d "
                                   "%s per GPU, one opt_min_curv pass (assembly + QP + "
                                   "curvature-error check) per track per step; kappa_bound=0.12, w_veh=3.4"
                                   % ("5" if args.perturb_centreline else "3", n, B,
                                      "reference tracks (centreline and widths perturbed per track)" if args.perturb_centreline
                                      else "track-width perturbations"),
                       "batch_per_gpu": B, "n_waypoints": n, "io": args.io + (" rows / alpha in HBM, fp64 arithmetic" if f32 else ""),
                       "centrelines": "perturbed per track" if args.perturb_centreline else "shared",
                       "collective": "1 all-gather of alpha per step" if collective else "none",
                       "failed_problems": n_bad,
                       "mean_ipm_iters": float(info["ipm_iters"].mean()), "mean_as_iters": float(info["as_iters"].mean()),
                       "mean_active_box_rows": float(info["n_active_box"].mean()),
                       "mean_refine_rounds": float(info["refine_rounds"].mean()),
                       "second_attempts": int(info["second_attempt"].sum()),
                       "kernel_ms": {k: float(np.mean([m[k] for m in solve_ms])) for k in ("assemble", "gram", "solve", "total")},
                       "workspace_GB": eng.workspace_bytes() / 1e9,
                       "solver_phase_ms_per_problem": {k: float(info["ticks"][:, j].mean()) / 1e5 for j, k in
                                                       enumerate(("factor", "solve", "gradient", "kernel", "sweep_fwd", "sweep_bwd"))}},
            "roofline": {"bound": "hbm", "kernel": "mcq_solve_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # the committed counter passes are of the default workload only
                         "traffic": measured_traffic() if (B == 1024 and n == 2000 and not args.perturb_centreline) else None,
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": k_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            if f32:     # the baseline solves the rows the engine saw
                ref_h = ref_h.astype(np.float32).astype(np.float64)
            a_cpu, err_cpu, t_cpu, k_cpu = cpu_baseline(ref_h, nv_h, sc_h)
            out["cpu_baseline"] = {"value": k_cpu / t_cpu, "unit": "solves/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "%d of the %d N=%d problems: dense-faithful numpy assembly (BLAS on all "
                                             "cores) + dense Goldfarb-Idnani in C (1 thread), %.1f s" % (k_cpu, B, n, t_cpu),
                                   "max_abs_alpha_diff_vs_gpu_m": float(np.max(np.abs(a_cpu - alpha0))),
                                   "curv_err_diff": abs(err_cpu - curv0)}
        print(json.dumps(out))
    if collective:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
