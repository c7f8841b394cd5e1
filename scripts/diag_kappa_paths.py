"""DIAGNOSTIC (round 5): curvature-constrained problems through the default path and through the Goldfarb-Idnani path alone: time per problem (kernel ticks) against the number of active curvature rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from global_racetrajectory_optimization_amd import engine
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "kappa_tight_fuzz.npz"))
off = z["offsets"]; nprob = len(off) - 1
probs = [dict(reftrack=z["reftrack"][off[k]:off[k + 1]], normvec=z["normvec"][off[k]:off[k + 1]], scaling=z["scaling"][off[k]:off[k + 1]],
              kappa_bound=float(z["kappa_bound"][k]), w_veh=float(z["w_veh"][k])) for k in range(nprob)]
eng = engine.Engine(0)
rows = []
for k in range(nprob):                      # one problem per launch: no contention for slots, clean per-problem times
    if z["status_ref"][k] != 0:
        continue
    _, _, s0, i0 = eng.solve_batch([probs[k]])
    _, _, s1, i1 = eng.solve_batch([probs[k]], algorithm=engine.ALG_GI)
    rows.append((int(z["n_active_kappa"][k]), int(z["n_active_box"][k]), probs[k]["reftrack"].shape[0], i0[0]["ticks"][3] / 1e5, i1[0]["ticks"][3] / 1e5,
                 i0[0]["gi_iters"], i1[0]["gi_iters"], i0[0]["as_iters"]))
rows.sort()
print("active curvature rows, box rows, n, default path ms, GI path ms, (fallback steps in default), GI steps, block-pivoting rounds")
for r in rows[::6] + rows[-12:]:
    print("%4d %4d %4d   %8.2f %8.2f   %4d %4d %3d" % r)
a = np.array([(r[0], r[3], r[4]) for r in rows])
for lo, hi in ((0, 0), (1, 10), (11, 40), (41, 120), (121, 1000)):
    m = (a[:, 0] >= lo) & (a[:, 0] <= hi)
    if m.any():
        print("active curvature rows %3d..%3d: %3d problems, default %.2f ms mean, GI %.2f ms mean" % (lo, hi, m.sum(), a[m, 1].mean(), a[m, 2].mean()))
