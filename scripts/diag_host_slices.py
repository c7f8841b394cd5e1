"""mcq_solve_host on 1024 x N = 2000 from pinned memory: one launch against 2 / 4 / 8 slices ($MCQ_HOST_SLICES), ms per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from global_racetrajectory_optimization_amd import engine, synthetic
B, n = 1024, 2000
ref, nv, sc = synthetic.oval_batch(B, n=n)
eng = engine.Engine(0)
p_ref, p_nv, p_sc, p_al = eng.host_array((B, n, 4)), eng.host_array((B, n, 2)), eng.host_array((B, n)), eng.host_array((B, n))
p_ref[...], p_nv[...], p_sc[...] = ref, nv, sc
base = None
for mode in ("one", "2", "4", "8", "one", "2"):
    os.environ.pop("MCQ_HOST_ONE_LAUNCH", None); os.environ.pop("MCQ_HOST_SLICES", None); 
    if mode == "one": os.environ["MCQ_HOST_ONE_LAUNCH"] = "1"
    else: os.environ["MCQ_HOST_SLICES"] = mode
    eng.solve_host(p_ref, p_nv, p_sc, 0.12, 3.4, alpha_out=p_al)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); eng.solve_host(p_ref, p_nv, p_sc, 0.12, 3.4, alpha_out=p_al); ts.append(time.perf_counter() - t)
    if base is None: base = p_al.copy()
    print("slices %s: %.2f ms (min %.2f), bitwise equal %s" % (mode, 1e3 * np.mean(ts), 1e3 * min(ts), np.array_equal(base, p_al)), flush=True)
