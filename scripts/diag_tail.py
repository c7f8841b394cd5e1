"""Per-problem spread behind the headline launch's tail: interior-point iteration counts and in-kernel times of the 1024 bench problems."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from global_racetrajectory_optimization_amd import engine, synthetic
import bench
B, n = 1024, 2000
ref, nv, sc = synthetic.oval_batch(B, n=n)
eng = engine.Engine(0)
d = dict(ref=eng.alloc(ref.nbytes), nv=eng.alloc(nv.nbytes), sc=eng.alloc(sc.nbytes), al=eng.alloc(8 * B * n), cu=eng.alloc(8 * B), st=eng.alloc(4 * B),
         info=eng.alloc(bench.INFO_DTYPE.itemsize * B))
eng.upload(d["ref"], ref); eng.upload(d["nv"], nv); eng.upload(d["sc"], sc)
for _ in range(3):
    eng.solve_device(B, n, d["ref"], d["nv"], d["sc"], 0.12, 3.4, d["al"], d["cu"], d["st"], d["info"])
eng.sync()
info = eng.download(d["info"], (B,), bench.INFO_DTYPE)
it = info["ipm_iters"]; ms = info["ticks"][:, 3] / 1e5
wd = np.stack([ref[:, :, 2].min(axis=1), ref[:, :, 2].mean(axis=1), (ref[:, :, 2] + ref[:, :, 3]).min(axis=1)])
print(json.dumps(dict(ipm_iters_hist={int(k): int(np.sum(it == k)) for k in np.unique(it)}, as_iters_hist={int(k): int(np.sum(info["as_iters"] == k)) for k in np.unique(info["as_iters"])},
                      kernel_ms=dict(mean=float(ms.mean()), std=float(ms.std()), p05=float(np.percentile(ms, 5)), p50=float(np.percentile(ms, 50)), p95=float(np.percentile(ms, 95)), max=float(ms.max())),
                      corr_ms_vs=dict(min_w_right=float(np.corrcoef(ms, wd[0])[0, 1]), mean_w_right=float(np.corrcoef(ms, wd[1])[0, 1]), min_corridor=float(np.corrcoef(ms, wd[2])[0, 1]),
                                      active_rows=float(np.corrcoef(ms, info["n_active_box"])[0, 1])))))
