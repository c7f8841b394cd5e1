#!/usr/bin/env python
"""mcq_iqp_batch on the 1024 synthetic ovals of BASELINE config 3, the first iters_min rounds one launch per round ($MCQ_IQP_FUSED=0) and
as one launch (mcq_iqp_rounds_kernel, the default): end to end seconds (best of three, end states into page-locked arrays), and the end
states of the second bitwise against those of the first.  One JSON line.

  python scripts/diag_iqp_modes.py [--batch 1024] [--n 2000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from global_racetrajectory_optimization_amd import engine, synthetic                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n", type=int, default=2000)
    args = ap.parse_args()
    eng = engine.Engine(0)
    ref, nv, sc = synthetic.oval_batch(args.batch, n=args.n)
    trk = dict(reftrack=ref, normvectors=nv, scaling=sc)
    os.environ["MCQ_IQP_FUSED"] = "0"
    w0 = eng.iqp_batch(trk, 0.12, 3.4, 3.0)
    nmx = w0["stats"]["nmax"]
    obuf = dict(alpha=eng.host_array((args.batch, nmx)), reftrack=eng.host_array((args.batch, nmx, 4)),
                normvectors=eng.host_array((args.batch, nmx, 2)))
    base = None
    rec = {}
    for mode, fused in (("one launch per round", "0"), ("one launch", "1")):
        os.environ["MCQ_IQP_FUSED"] = fused
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            r = eng.iqp_batch(trk, 0.12, 3.4, 3.0, iters_min=3, curv_error_allowed=0.01, nmax=nmx, out=obuf)
            ts.append(time.perf_counter() - t0)
        state = dict(alpha=[a.copy() for a in r["alpha"]], reftrack=[a.copy() for a in r["reftrack"]], n=r["n"].copy(),
                     status=r["status"].copy(), rounds=r["rounds"].copy(), curv=r["curv_err"].copy())
        if base is None:
            base = state
        same = (np.array_equal(state["n"], base["n"]) and np.array_equal(state["status"], base["status"]) and
                np.array_equal(state["rounds"], base["rounds"]) and np.array_equal(state["curv"], base["curv"]) and
                all(np.array_equal(a, b) for a, b in zip(state["alpha"], base["alpha"])) and
                all(np.array_equal(a, b) for a, b in zip(state["reftrack"], base["reftrack"])))
        rec[mode] = {"seconds": [round(t, 5) for t in ts], "qp_solves_per_s": r["stats"]["qp_solves"] / min(ts),
                     "rounds": r["stats"]["rounds"], "failed": int(np.count_nonzero(r["status"])), "bitwise_equal_to_the_loop": bool(same)}
    print(json.dumps({"what": "mcq_iqp_batch, %d tracks, N = %d" % (args.batch, args.n), "by_mode": rec}))
    eng.close()


if __name__ == "__main__":
    main()
