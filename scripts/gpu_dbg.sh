#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for so in global_racetrajectory_optimization_amd/csrc/variants/*.so; do
  name=$(basename $so .so | sed 's/^libmcq_//')
  for i in 1 2 3 4; do
  MCQ_LIB=$R/$so python - <<PY
import sys, numpy as np
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from global_racetrajectory_optimization_amd import engine
from conftest import load_golden, TRACKS
eng = engine.Engine(0)
g = {k: load_golden(k) for k in TRACKS}
probs = [dict(reftrack=g[k]["reftrack"], normvec=g[k]["normvec"], scaling=g[k]["scaling"], kappa_bound=0.12, w_veh=3.4) for k in TRACKS]
al, curv, st, info = eng.solve_batch(probs)
print("$name run $i: st", list(st), "tk7", [int(info[k]["ticks"][7]) for k in range(4)], "ipm", [info[k]["ipm_iters"] for k in range(4)])
for k in range(0):
    if st[k] != 0 or info[k]["ticks"][7] != 0: print("$name run $i:", TRACKS[k], "status", st[k], "ipm", info[k]["ipm_iters"], "as", info[k]["as_iters"], "tk7", info[k]["ticks"][7])
PY
  done
done
echo done
