"""A/B of the warm-started exchange's single-pivot switch (MCQ_AS_SINGLE_BELOW builds): the bench's IQP workload, per-pass solver ms, fallbacks, end to end."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from global_racetrajectory_optimization_amd import engine, synthetic
B, n = 1024, 2000
ref, nv, sc = synthetic.oval_batch(B, n=n)
trk = dict(reftrack=ref, normvectors=nv, scaling=sc)
base = None
for lib in sys.argv[1:]:
    eng = engine.Engine(0, lib_path=None if lib == "default" else lib)
    w0 = eng.iqp_batch(trk, 0.12, 3.4, 3.0)
    nmx = w0["stats"]["nmax"]
    obuf = dict(alpha=eng.host_array((B, nmx)), reftrack=eng.host_array((B, nmx, 4)), normvectors=eng.host_array((B, nmx, 2)))
    ts = []
    for _ in range(3):
        t = time.perf_counter(); iq = eng.iqp_batch(trk, 0.12, 3.4, 3.0, iters_min=3, curv_error_allowed=0.01, nmax=nmx, out=obuf); ts.append(time.perf_counter() - t)
    iqt = eng.iqp_batch(trk, 0.12, 3.4, 3.0, iters_min=3, curv_error_allowed=0.01, timed=True, nmax=nmx)
    al = np.concatenate([a for a in iq["alpha"]])
    if base is None: base = al
    print(json.dumps(dict(lib=os.path.basename(lib), seconds=min(ts), qp_per_s=iq["stats"]["qp_solves"] / min(ts), pass_ms=iqt["stats"]["solver_ms"], fallbacks=iqt["stats"]["fallbacks"],
                          failed=int(np.count_nonzero(iq["status"])), max_abs_diff_vs_first=float(np.max(np.abs(al - base))) if al.shape == base.shape else None)), flush=True)
    eng.close()
