#!/bin/bash
# runs every build/fb/fb_* (scripts/factor_bench.hip variants) on the GPU box: cycles per factorisation step of each
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-fb}
mkdir -p $R/gpurun_out
cd $R
for f in build/fb/fb_*; do
  for fw in 0 1; do
    timeout 120 $f 1024 2000 12 $fw 2>&1 | python3 -c "
import sys, json
name, fw = sys.argv[1], sys.argv[2]
for line in sys.stdin:
    try:
        d = json.loads(line); print('%-10s fwd%s abl %5d  cycles/step %6.0f  us/fact %7.2f  notpd %d' % (name, fw, d['abl'], d['cycles_per_step'], d['us_per_factorisation_per_workgroup'], d['not_pd']))
    except Exception: print(name, line.strip())
" $(basename $f) $fw
  done
done | tee gpurun_out/${T}_factor_bench.txt
