#!/bin/bash
# Repeats one quick parity check many times per build variant (race hunting).  scripts/gpu_flaky.sh <reps>
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-30}
cd $R
for so in global_racetrajectory_optimization_amd/csrc/variants/*.so; do
  name=$(basename $so .so | sed 's/^libmcq_//')
  MCQ_LIB=$R/$so python - <<PY
import sys, numpy as np
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from global_racetrajectory_optimization_amd import engine
from conftest import load_golden, TRACKS
eng = engine.Engine(0)
g = {k: load_golden(k) for k in TRACKS}
probs = [dict(reftrack=g[k]["reftrack"], normvec=g[k]["normvec"], scaling=g[k]["scaling"], kappa_bound=0.12, w_veh=3.4) for k in TRACKS]
bad = 0; worst = 0.0
for rep in range($N):
    mult = (1, 1, 1, 2, 8, 64)[rep % 6]
    al, curv, st, info = eng.solve_batch(probs * mult)
    for k in range(len(st)):
        name = TRACKS[k % 4]
        if st[k] != 0: bad += 1
        else: worst = max(worst, float(np.max(np.abs(al[k] - g[name]["alpha"]))))
print("$name: %d reps x (4 | 8 | 32 | 256) problems: %d non-zero status, worst |d alpha| of the rest %.2e" % ($N, bad, worst))
PY
done
