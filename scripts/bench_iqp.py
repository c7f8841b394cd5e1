#!/usr/bin/env python
"""Measurement for row f-1 (device-side IQP glue): time of mcq_relinearise_device on a batch of N = 2000 rings, and the
throughput of whole IQP runs (iters_min = 3) with the glue on the device vs on the host.  Prints one JSON line.

  python scripts/bench_iqp.py [--batch 1024] [--n 2000] [--iqp-batch 64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from global_racetrajectory_optimization_amd import engine, synthetic                                  # noqa: E402
from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--iqp-batch", type=int, default=64)
    ap.add_argument("--no-host", action="store_true", help="skip the host-glue run (slow for large batches)")
    args = ap.parse_args()
    eng = engine.Engine(0)
    B, n = args.batch, args.n
    ref, nv, sc = synthetic.oval_batch(B, n=n, first=0)
    nmax = n + 128
    ref_p = np.zeros((B, nmax, 4)); ref_p[:, :n] = ref
    nv_p = np.zeros((B, nmax, 2)); nv_p[:, :n] = nv
    alpha = 0.5 * np.sin(np.arange(nmax) * 0.01)[None, :] * np.ones((B, 1))
    f8, i4 = 8, 4
    d_ref = eng.alloc(ref_p.nbytes); eng.upload(d_ref, ref_p)
    d_nv = eng.alloc(nv_p.nbytes); eng.upload(d_nv, nv_p)
    d_al = eng.alloc(alpha.nbytes); eng.upload(d_al, alpha)
    d_n = eng.alloc(B * i4); eng.upload(d_n, np.full(B, n, dtype=np.int32))
    d_ref2 = eng.alloc(ref_p.nbytes); d_nv2 = eng.alloc(nv_p.nbytes); d_n2 = eng.alloc(B * i4); d_st = eng.alloc(B * i4)
    # the relinearise kernel borrows the solver's vector slab: size the workspace with one (tiny) solve first
    d_a1 = eng.alloc(B * nmax * f8); d_c1 = eng.alloc(B * f8); d_s1 = eng.alloc(B * i4)

    def relin():
        eng.relinearise_device(B, nmax, d_n, d_ref, d_nv, d_al, None, 1.0, 3.0, d_ref2, d_nv2, d_n2, d_st)
    relin(); eng.sync()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        relin()
    eng.sync()
    t_relin = (time.perf_counter() - t0) / reps
    st = eng.download(d_st, (B,), np.int32)
    n2 = eng.download(d_n2, (B,), np.int32)
    # bytes the glue must move per track: read ref (4N) + nv (2N) + alpha (N), write ref' (4M) + nv' (2M)
    alg_bytes = B * (7 * n + 6 * float(n2.mean())) * f8

    Bi = args.iqp_batch
    tracks = [dict(reftrack=ref[k].copy(), normvectors=nv[k], scaling=sc[k]) for k in range(Bi)]
    res = {}
    for mode in ((True,) if args.no_host else (True, False)):
        first = None
        for rep in range(2 if mode else 1):     # device mode: the first run also pays for growing the engine's workspace
            stats = {}
            t0 = time.perf_counter()
            out = iqp_handler.iqp_handler_batch([dict(t, reftrack=t["reftrack"].copy()) for t in tracks], 0.12, 3.4, 3.0, 3, 0.01,
                                                engine=eng, stats=stats, device_resident=mode)
            dt = time.perf_counter() - t0
            first = dt if first is None else first
        res["device" if mode else "host"] = dict(seconds=dt, seconds_first_run=first, qp_solves=stats["qp_solves"],
                                                  rounds=stats["rounds"], qp_solves_per_s=stats["qp_solves"] / dt,
                                                  n_last=int(out[0][0].shape[0]), alpha0=out[0][0],
                                                  **{k: stats[k] for k in ("seconds_upload", "seconds_rounds", "seconds_download") if k in stats})
    diff = None
    if "host" in res and res["device"]["n_last"] == res["host"]["n_last"]:
        diff = float(np.max(np.abs(res["device"]["alpha0"] - res["host"]["alpha0"])))
    for r in res.values():
        del r["alpha0"]
    print(json.dumps({"relinearise": {"batch": B, "n": n, "ms": 1e3 * t_relin, "status_ok": bool(np.all(st == 0)),
                                      "n_out_mean": float(n2.mean()), "algorithmic_GB": alg_bytes / 1e9,
                                      "achieved_GBs": alg_bytes / t_relin / 1e9},
                      "iqp": {"tracks": Bi, "n": n, **res, "max_abs_alpha_diff_device_vs_host": diff}}))


if __name__ == "__main__":
    main()
