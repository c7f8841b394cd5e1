#!/bin/bash
# LDS / workspace poison check (MCQ_POISON=1: every allocation and the solver kernel's LDS start out as NaN patterns) per build variant,
# then the whole GPU suite on the product library with and without the poison.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1
for so in global_racetrajectory_optimization_amd/csrc/variants/*.so; do
  for poison in 0 1; do
  name=$(basename $so .so | sed 's/^libmcq_//')
  MCQ_POISON=$poison MCQ_LIB=$R/$so NAME="$name poison=$poison" python - <<PY
import os, sys, numpy as np
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
from global_racetrajectory_optimization_amd import engine
from conftest import load_golden, TRACKS
eng = engine.Engine(0)
g = {k: load_golden(k) for k in TRACKS}
probs = [dict(reftrack=g[k]["reftrack"], normvec=g[k]["normvec"], scaling=g[k]["scaling"], kappa_bound=0.12, w_veh=3.4) for k in TRACKS] * 16
bad = 0
for r in range(12):
    al, curv, st, info = eng.solve_batch(probs)
    bad += int(any(int(s) != 0 for s in st))
print(os.environ["NAME"], "bad runs:", bad, "of 12")
PY
  done
done
MCQ_POISON=1 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
