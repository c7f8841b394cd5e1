#!/usr/bin/env python
"""Per-pass statistics of a batch of device-resident IQP runs on the synthetic ovals (BASELINE config 3): solver-kernel time,
interior-point / active-set iterations, second attempts, phase times per problem.  One JSON line per pass.

  python scripts/diag_iqp_rounds.py [--batch 1024] [--n 2000] [--passes 3]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from global_racetrajectory_optimization_amd import engine, synthetic                      # noqa: E402

INFO_DTYPE = np.dtype([("ipm_iters", "<i4"), ("as_iters", "<i4"), ("n_active_box", "<i4"), ("n_active_kappa", "<i4"),
                       ("kappa_max", "<f8"), ("kkt_res", "<f8"), ("ticks", "<i8", (8,)),
                       ("refine_rounds", "<i4"), ("second_attempt", "<i4"), ("f32_factorisations", "<i4"), ("gi_iters", "<i4")])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--iters-min", type=int, default=3)
    ap.add_argument("--cold", action="store_true", help="no warm start of passes 2+")
    ap.add_argument("--dump", default=None, help="write every pass's per-track mcq_info records to this .npz")
    args = ap.parse_args()
    assert INFO_DTYPE.itemsize == ctypes.sizeof(engine.McqInfo)
    eng = engine.Engine(0)
    bsz, n = args.batch, args.n
    ref, nv, sc = synthetic.oval_batch(bsz, n=n)
    nmax = int(1.3 * n) + 16
    ref_h = np.zeros((bsz, nmax, 4)); ref_h[:, :n] = ref
    nv_h = np.zeros((bsz, nmax, 2)); nv_h[:, :n] = nv
    sc_h = np.ones((bsz, nmax)); sc_h[:, :n] = sc
    n_h = np.full(bsz, n, dtype=np.int32)

    def up(a):
        p = eng.alloc(a.nbytes)
        eng.upload(p, a)
        return p
    d_ref = [up(ref_h), eng.alloc(ref_h.nbytes)]
    d_nv = [up(nv_h), eng.alloc(nv_h.nbytes)]
    d_sc = up(sc_h)
    d_n = [up(n_h), eng.alloc(bsz * 4)]
    d_alpha, d_curv, d_status = eng.alloc(bsz * nmax * 8), eng.alloc(bsz * 8), eng.alloc(bsz * 4)
    d_info = eng.alloc(bsz * INFO_DTYPE.itemsize)
    d_rst = eng.alloc(bsz * 4)
    cur = 0
    eng.prep_batch([ref_h[0, :n]] * 2)          # HIP module load
    dump = {}
    for it in range(1, args.passes + 1):
        # (a warm start is consumed by the launch that uses it: one timed launch per pass; the workspace is grown beforehand)
        eng.solve_device_ragged(bsz, nmax, d_n[cur], d_ref[cur], d_nv[cur], d_sc if it == 1 else None, 0.12, 3.4, d_alpha,
                                d_curv, d_status, d_info, warm_start=0 if (args.cold or it == 1) else 1)
        eng.sync()
        ms = eng.last_timing_ms()
        info = eng.download(d_info, (bsz * INFO_DTYPE.itemsize,), np.uint8).view(INFO_DTYPE)
        status = eng.download(d_status, (bsz,), np.int32)
        curv = eng.download(d_curv, (bsz,), np.float64)
        t = info["ticks"].astype(np.float64) / 1e5
        for key in ("ipm_iters", "as_iters", "n_active_box", "second_attempt"):
            dump["pass%d_%s" % (it, key)] = np.array(info[key])
        dump["pass%d_kernel_ms" % it] = t[:, 3].copy()
        print(json.dumps({"pass": it, "kernel_ms": ms, "status_nonzero": int(np.count_nonzero(status)),
                          "curv_err_max": float(curv.max()), "curv_err_mean": float(curv.mean()),
                          "ipm_iters": [float(info["ipm_iters"].mean()), int(info["ipm_iters"].max())],
                          "as_iters": [float(info["as_iters"].mean()), int(info["as_iters"].max())],
                          "active_box": [float(info["n_active_box"].mean()), int(info["n_active_box"].max())],
                          "active_kappa_max": int(info["n_active_kappa"].max()),
                          "as_hist": np.bincount(np.minimum(info["as_iters"], 20), minlength=21).tolist(),
                          "worst": [[int(k), int(info["as_iters"][k]), int(info["ipm_iters"][k]), float(t[k, 3])]
                                    for k in np.argsort(-t[:, 3])[:8]],
                          "second_attempts": int(info["second_attempt"].sum()), "refine_rounds": float(info["refine_rounds"].mean()),
                          "ms_per_problem": {"factor": float(t[:, 0].mean()), "solve": float(t[:, 1].mean()), "gradient": float(t[:, 2].mean()),
                                             "kernel_mean": float(t[:, 3].mean()), "kernel_max": float(t[:, 3].max()),
                                             "kernel_p90": float(np.percentile(t[:, 3], 90))}}))
        scale = it / args.iters_min if it < args.iters_min else 1.0
        eng.relinearise_device(bsz, nmax, d_n[cur], d_ref[cur], d_nv[cur], d_alpha, None, scale, 3.0, d_ref[1 - cur], d_nv[1 - cur],
                               d_n[1 - cur], d_rst)
        eng.sync()
        cur = 1 - cur
    if args.dump:
        np.savez_compressed(args.dump, **dump)
    eng.close()


if __name__ == "__main__":
    main()
