#!/bin/bash
# Parity check repeated while a second process saturates HBM (timing skew between the waves of a workgroup: race hunting)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
python - <<'PY' &
import torch, time
a = torch.empty(2 << 28, dtype=torch.float64, device="cuda")   # 4 GiB
b = torch.empty_like(a)
t0 = time.time()
while time.time() - t0 < 150:
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize()
PY
HOG=$!
sleep 8
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
bad = 0; worst = 0.0; tot = 0
for rep in range(${1:-60}):
    mult = (1, 1, 2, 8, 64, 256)[rep % 6]
    al, curv, st, info = eng.solve_batch(probs * mult)
    tot += len(st)
    for k in range(len(st)):
        if st[k] != 0: bad += 1
        else: worst = max(worst, float(np.max(np.abs(al[k] - g[TRACKS[k % 4]]["alpha"]))))
print("$name under HBM load: %d solves, %d non-zero status, worst |d alpha| of the rest %.2e" % (tot, bad, worst))
PY
done
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
