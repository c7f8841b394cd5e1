import sys, ctypes, numpy as np
sys.path.insert(0, ".")
from global_racetrajectory_optimization_amd import engine, synthetic
eng = engine.Engine(0)
ref, nv, sc = synthetic.oval_batch(1024, n=2000)
probs = [dict(reftrack=ref[k], normvec=nv[k], scaling=sc[k], kappa_bound=0.12, w_veh=3.4) for k in range(1024)]
al, curv, st, info = eng.solve_batch(probs)
t = np.array([i["ticks"][3] for i in info], float) / 1e5
ipm = np.array([i["ipm_iters"] for i in info], float); asi = np.array([i["as_iters"] for i in info], float); act = np.array([i["n_active_box"] for i in info], float)
wmin = np.array([(r[:, 2] + r[:, 3]).min() for r in ref]); wmean = np.array([(r[:, 2] + r[:, 3]).mean() for r in ref])
for name, v in (("ipm", ipm), ("as", asi), ("active", act), ("wmin", wmin), ("wmean", wmean)):
    print(name, "corr with time %.3f" % np.corrcoef(v, t)[0, 1])
print("time mean %.2f std %.2f min %.2f max %.2f" % (t.mean(), t.std(), t.min(), t.max()))
print("ipm hist", np.bincount(ipm.astype(int)))
# LPT simulation: 256 servers, dispatch in given order
def makespan(order):
    import heapq
    h = [0.0] * 256
    heapq.heapify(h)
    for k in order:
        s = heapq.heappop(h); heapq.heappush(h, s + t[k])
    return max(h)
print("makespan index order %.2f  LPT(perfect) %.2f  by wmean asc %.2f  by wmin asc %.2f mean load %.2f" % (makespan(range(1024)), makespan(np.argsort(-t)), makespan(np.argsort(wmean)), makespan(np.argsort(wmin)), t.sum() / 256))
