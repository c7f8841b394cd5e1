import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from global_racetrajectory_optimization_amd import engine
lib = sys.argv[1] if len(sys.argv) > 1 else None
eng = engine.Engine(0, lib_path=lib)
for name in ("rounded_rectangle", "handling_track", "berlin_2018"):
    g = np.load("tests/golden/%s.npz" % name)
    t = time.time()
    al, cu, st, info = eng.solve_batch([dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12, w_veh=3.4)], algorithm=engine.ALG_GI)
    print(lib, name, "status", st[0], "gi_iters", info[0]["gi_iters"], "diff %.2e" % np.max(np.abs(al[0] - g["alpha"])), "%.3fs" % (time.time() - t), flush=True)
