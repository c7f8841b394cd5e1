#!/usr/bin/env python
"""Diagnostic: the shortest-path QP of bench.py's 1024 ovals -- the engine's self-reported KKT residual against a certificate computed on the
host from H, f and the box; twice (determinism)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from global_racetrajectory_optimization_amd import engine, synthetic

def cert(ref, nv, w_veh, x):
    p = ref[:, :2]
    hd = 4.0 * np.sum(nv * nv, axis=1)
    hu = -2.0 * np.sum(nv * np.roll(nv, -1, axis=0), axis=1)
    f = 2.0 * np.sum(nv * (2.0 * p - np.roll(p, 1, axis=0) - np.roll(p, -1, axis=0)), axis=1)
    lo, hi = -np.maximum(ref[:, 3] - w_veh / 2, 0.001), np.maximum(ref[:, 2] - w_veh / 2, 0.001)
    g = hd * x + hu * np.roll(x, -1) + np.roll(hu, 1) * np.roll(x, 1) + f
    free = (x > lo + 1e-9) & (x < hi - 1e-9)
    v = np.max(np.abs(g[free])) if free.any() else 0.0
    v = max(v, np.max(np.maximum(-g[x <= lo + 1e-9], 0.0), initial=0.0), np.max(np.maximum(g[x >= hi - 1e-9], 0.0), initial=0.0))
    return v / np.max(np.abs(f)), float(np.max(np.maximum(lo - x, x - hi)))

B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 2000
eng = engine.Engine(0)
ref, nv, _ = synthetic.oval_batch(B, n=n)
probs = [dict(reftrack=ref[k], normvec=nv[k], scaling=None, kappa_bound=1.0, w_veh=3.4) for k in range(B)]
prev = None
for rep in range(2):
    al, curv, st, info = eng.solve_batch(probs, objective=engine.OBJ_SHORTEST_PATH)
    kk = np.array([i["kkt_res"] for i in info])
    bad = np.nonzero(~(kk < 1e-9))[0]
    cs = [cert(ref[k], nv[k], 3.4, al[k]) for k in range(B)]
    worst = max(c[0] for c in cs)
    print("rep", rep, "status != 0:", int(np.count_nonzero(st)), "self-reported kkt > 1e-9:", bad.size, bad[:10].tolist(), [float(kk[b]) for b in bad[:5]],
          "host certificate: worst violation %.2e, worst infeasibility %.2e" % (worst, max(c[1] for c in cs)),
          "of the flagged: %s" % [("%.1e" % cs[b][0]) for b in bad[:5]], "refine", [info[b]["refine_rounds"] for b in bad[:5]], "as", [info[b]["as_iters"] for b in bad[:5]])
    if prev is not None:
        print("bitwise equal to the first run:", all(np.array_equal(a, b) for a, b in zip(al, prev)))
    prev = al
