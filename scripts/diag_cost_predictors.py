"""DIAGNOSTIC: how much of a launch's tail (the launch ends with its slowest problems) an ordering of the batch could take back.  Per-problem
kernel times from mcq_info.ticks of one launch of the bench workload; list-scheduling simulation on 512 slots (two workgroups per CU; the
two of a CU share its bandwidth, which the simulation ignores) for the index order, the perfect longest-first order and a few cheap
predictors."""
import sys, heapq
import numpy as np
sys.path.insert(0, ".")
from global_racetrajectory_optimization_amd import engine, synthetic
eng = engine.Engine(0)
ref, nv, sc = synthetic.oval_batch(1024, n=2000)
probs = [dict(reftrack=ref[k], normvec=nv[k], scaling=sc[k], kappa_bound=0.12, w_veh=3.4) for k in range(1024)]
al, curv, st, info = eng.solve_batch(probs)
al, curv, st, info = eng.solve_batch(probs)
t = np.array([i["ticks"][3] for i in info], float) / 1e5          # ms
ipm = np.array([i["ipm_iters"] for i in info], float)
asi = np.array([i["as_iters"] for i in info], float)
act = np.array([i["n_active_box"] for i in info], float)
wmin = np.array([(r[:, 2] + r[:, 3]).min() for r in ref])
wmean = np.array([(r[:, 2] + r[:, 3]).mean() for r in ref])
narrow = np.array([np.count_nonzero((r[:, 2] + r[:, 3]) < 3.4 + 1.0) for r in ref], float)
for name, v in (("ipm", ipm), ("as", asi), ("active", act), ("wmin", wmin), ("wmean", wmean), ("narrow", narrow)):
    print(name, "corr with time %.3f" % np.corrcoef(v, t)[0, 1])
print("time mean %.3f std %.3f min %.3f max %.3f ms" % (t.mean(), t.std(), t.min(), t.max()))
print("ipm hist", np.bincount(ipm.astype(int)), "as hist", np.bincount(asi.astype(int)))


def makespan(order, slots=512):
    h = [0.0] * slots
    heapq.heapify(h)
    for k in order:
        s = heapq.heappop(h)
        heapq.heappush(h, s + t[k])
    return max(h)


print("makespan: index order %.2f  longest first (perfect knowledge) %.2f  by active rows desc %.2f  by wmean asc %.2f  mean load %.2f ms"
      % (makespan(range(1024)), makespan(np.argsort(-t)), makespan(np.argsort(-act)), makespan(np.argsort(wmean)), t.sum() / 512))
eng.close()
