#!/usr/bin/env python
"""Host-side time stamps of one mcq_iqp_batch call on the 1024 ovals ($MCQ_IQP_TRACE: stderr), with the Python binding's share around it."""
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from global_racetrajectory_optimization_amd import engine, synthetic                      # noqa: E402

eng = engine.Engine(0)
B, n = 1024, 2000
ref, nv, sc = synthetic.oval_batch(B, n=n)
trk = dict(reftrack=ref, normvectors=nv, scaling=sc)
w0 = eng.iqp_batch(trk, 0.12, 3.4, 3.0)
nmx = w0["stats"]["nmax"]
obuf = dict(alpha=eng.host_array((B, nmx)), reftrack=eng.host_array((B, nmx, 4)), normvectors=eng.host_array((B, nmx, 2)))
eng.iqp_batch(trk, 0.12, 3.4, 3.0, nmax=nmx, out=obuf)
os.environ["MCQ_IQP_TRACE"] = "1"
for rep in range(2):
    t0 = time.perf_counter()
    eng.iqp_batch(trk, 0.12, 3.4, 3.0, nmax=nmx, out=obuf)
    sys.stderr.write("python: whole call %.3f ms\n" % (1e3 * (time.perf_counter() - t0)))
eng.close()
