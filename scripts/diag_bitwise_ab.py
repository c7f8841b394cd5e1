"""DIAGNOSTIC: do two builds of the library return the same alpha bit for bit on the bench workload?  python scripts/diag_bitwise_ab.py libA.so libB.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from global_racetrajectory_optimization_amd import engine, synthetic
ref, nv, sc = synthetic.oval_batch(256, n=2000)
out = []
for lib in sys.argv[1:3]:
    eng = engine.Engine(0, lib_path=lib)
    a, c, s, _ = eng.solve_host(ref, nv, sc, 0.12, 3.4)
    out.append((a.copy(), c.copy(), s.copy()))
    eng.close()
print("alpha bitwise equal:", bool(np.array_equal(out[0][0], out[1][0])), "max diff %.2e" % float(np.max(np.abs(out[0][0] - out[1][0]))), "status", np.unique(out[0][2]), np.unique(out[1][2]))
