"""The bench workload (1024 ovals, N = 2000) with EVERY problem through the Goldfarb-Idnani path (mcq_opts.algorithm = MCQ_ALG_GI): quadprog's algorithm
on the GPU, a slot per resident workgroup.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from global_racetrajectory_optimization_amd import engine, synthetic
B, n = int(os.environ.get("GI_BATCH", "1024")), 2000
ref, nv, sc = synthetic.oval_batch(B, n=n)
eng = engine.Engine(0)
d = dict(ref=eng.alloc(ref.nbytes), nv=eng.alloc(nv.nbytes), sc=eng.alloc(sc.nbytes), al=eng.alloc(8 * B * n), cu=eng.alloc(8 * B), st=eng.alloc(4 * B))
eng.upload(d["ref"], ref); eng.upload(d["nv"], nv); eng.upload(d["sc"], sc)
eng.solve_device(B, n, d["ref"], d["nv"], d["sc"], 0.12, 3.4, d["al"], d["cu"], d["st"])
eng.sync()
a0 = eng.download(d["al"], (B, n), np.float64)
ts = []
for rep in range(3):
    eng.sync(); t0 = time.perf_counter()
    eng.solve_device(B, n, d["ref"], d["nv"], d["sc"], 0.12, 3.4, d["al"], d["cu"], d["st"], algorithm=engine.ALG_GI)
    eng.sync(); ts.append(time.perf_counter() - t0)
a1 = eng.download(d["al"], (B, n), np.float64); st = eng.download(d["st"], (B,), np.int32)
print(json.dumps({"what": "every problem through the Goldfarb-Idnani path (mcq_opts.algorithm = MCQ_ALG_GI)", "batch": B, "n": n, "seconds": ts,
                  "solves_per_s": B / min(ts), "failed": int(np.count_nonzero(st)), "max_abs_alpha_diff_vs_default_path_m": float(np.max(np.abs(a1 - a0))),
                  "workspace_GB": eng.workspace_bytes() / 1e9}))
