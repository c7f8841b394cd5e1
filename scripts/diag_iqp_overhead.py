#!/usr/bin/env python
"""Where the end-to-end time of mcq_iqp_batch goes outside the QP passes (host packing, PCIe, page faults of fresh output arrays)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from global_racetrajectory_optimization_amd import engine, synthetic

eng = engine.Engine(0)
ref, nv, sc = synthetic.oval_batch(1024, n=2000)
tracks = [dict(reftrack=ref[k], normvectors=nv[k], scaling=sc[k]) for k in range(1024)]
for rep in range(3):
    t0 = time.perf_counter()
    out = eng.iqp_batch(tracks, 0.12, 3.4, 3.0, 3, 0.01, timed=True)
    t1 = time.perf_counter()
    print("iqp_batch %.1f ms, solver passes %s (sum %.1f), nmax %d" % ((t1 - t0) * 1e3, [round(x, 1) for x in out["stats"]["solver_ms"]], sum(out["stats"]["solver_ms"]), out["stats"]["nmax"]))
nmax = out["stats"]["nmax"]
# raw copies
d = eng.alloc(1024 * nmax * 4 * 8)
for name, mk in (("pageable fresh", lambda: np.zeros((1024, nmax, 4))), ("pageable touched", lambda: np.ones((1024, nmax, 4))), ("pinned", lambda: eng.host_array((1024, nmax, 4)))):
    a = mk()
    t0 = time.perf_counter(); eng.lib.mcq_copy_to_host(eng.h, a.ctypes.data, d, a.nbytes); eng.sync(); t1 = time.perf_counter()
    print("D2H %d MB into %s: %.1f ms" % (a.nbytes >> 20, name, (t1 - t0) * 1e3))
t0 = time.perf_counter(); z = np.zeros((1024, nmax, 7)); t1 = time.perf_counter(); z[...] = 1.0; t2 = time.perf_counter()
print("np.zeros 150 MB %.1f ms, first touch %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
src = np.ones((1024, 2000, 7)); dst = eng.host_array((1024, 2000, 7))
t0 = time.perf_counter(); dst[...] = src; t1 = time.perf_counter()
print("host memcpy 115 MB pageable -> pinned: %.1f ms" % ((t1 - t0) * 1e3))
