#!/usr/bin/env python
"""Measurement for row f-4: the shortest-path QP on bench.py's workload (1024 perturbed ovals, N = 2000), inputs resident
in HBM, device events around the kernels.  One JSON line.

  python scripts/bench_shortest_path.py [--batch 1024] [--n 2000] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import INFO_DTYPE                                                     # noqa: E402
from global_racetrajectory_optimization_amd import engine, synthetic            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    B, n = args.batch, args.n
    eng = engine.Engine(0)
    ref, nv, _ = synthetic.oval_batch(B, n=n)
    d_ref, d_nv = eng.alloc(ref.nbytes), eng.alloc(nv.nbytes)
    d_alpha, d_curv, d_st, d_info = eng.alloc(8 * B * n), eng.alloc(8 * B), eng.alloc(4 * B), eng.alloc(INFO_DTYPE.itemsize * B)
    eng.upload(d_ref, ref)
    eng.upload(d_nv, nv)
    ms = []
    t0 = None
    for k in range(args.steps + 1):
        if k == 1:
            eng.sync()
            t0 = time.perf_counter()
        eng.solve_device(B, n, d_ref, d_nv, None, 1.0, 3.4, d_alpha, d_curv, d_st, d_info,
                         objective=engine.OBJ_SHORTEST_PATH)
        eng.sync()
        if k:
            ms.append(eng.last_timing_ms())
    dt = time.perf_counter() - t0
    st = eng.download(d_st, (B,), np.int32)
    info = eng.download(d_info, (B,), INFO_DTYPE)
    print(json.dumps({"objective": "shortest_path", "batch": B, "n": n, "steps": args.steps,
                      "solves_per_s": B * args.steps / dt, "failed": int(np.count_nonzero(st)),
                      "kernel_ms": {k: float(np.mean([m[k] for m in ms])) for k in ("assemble_sp", "solve", "total")},
                      "mean_ipm_iters": float(info["ipm_iters"].mean()), "mean_as_iters": float(info["as_iters"].mean()),
                      "max_as_iters": int(info["as_iters"].max()), "second_attempts": int(info["second_attempt"].sum()),
                      "mean_active_box_rows": float(info["n_active_box"].mean()),
                      "max_kkt_res": float(info["kkt_res"].max())}))
    for p in (d_ref, d_nv, d_alpha, d_curv, d_st, d_info):
        eng.free(p)
    eng.close()


if __name__ == "__main__":
    main()
