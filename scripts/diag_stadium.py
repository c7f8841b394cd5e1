"""DIAGNOSTIC: the 720-point stadium with ~270 active curvature rows through the library named by MCQ_LIB; prints status and counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from global_racetrajectory_optimization_amd import engine
from test_emu_kernels import stadium_problem
eng = engine.Engine(0)
for n in (360, 720):
    ref, nv, A, sc, kb = stadium_problem(n, 0.0223)
    al, curv, st, inf = eng.solve_batch([dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=kb, w_veh=2.0)])
    print(os.environ.get("MCQ_LIB", "default"), n, "status", int(st[0]), "curv_err %.6e" % curv[0], inf[0], "alpha checksum %.12e" % float(np.sum(al[0] * np.arange(1, n + 1))))
eng.close()
