#!/bin/bash
# Builds same-box A/B variants of libmcq.so into build/variants/ (git-ignored; shipped to the GPU box by gpurun):
#   scripts/build_variants.sh name1 "-DFLAG=1 ..." name2 "..." ...
# `00base` is always built first: the kernels of the commit given by BASE_REV (default: the previous round's last commit) with THIS
# tree's C ABI around them, so that every variant loads through the same engine.py.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/global_racetrajectory_optimization_amd/csrc
OUT=$R/build/variants
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared -mllvm -enable-ipra=0 -mllvm -amdgpu-schedule-metric-bias=0 --gpu-max-threads-per-block=512"
mkdir -p $OUT /tmp/mcq_base/csrc /tmp/mcq_base/include_dir
rm -f $OUT/*.so
BASE_REV=${BASE_REV:-3b86c49}
if [ "$BASE_REV" != "none" ]; then
  # the whole csrc of BASE_REV (its own C ABI: entries added since are missing -- variants scripts only call what round 2 had)
  rm -rf /tmp/mcq_base && mkdir -p /tmp/mcq_base/csrc /tmp/mcq_base/include
  for f in $(git -C $R ls-tree --name-only $BASE_REV:global_racetrajectory_optimization_amd/csrc | grep -E "\.(hip|h|inc)$"); do git -C $R show $BASE_REV:global_racetrajectory_optimization_amd/csrc/$f > /tmp/mcq_base/csrc/$f; done
  git -C $R show $BASE_REV:include/mcq.h > /tmp/mcq_base/include/mcq.h
  sed -i 's#"../../include/mcq.h"#"../include/mcq.h"#' /tmp/mcq_base/csrc/mcq_kernels.h
  (cd /tmp/mcq_base/csrc && $HIPCC $FLAGS -o $OUT/libmcq_00base.so mcq_kernels.hip mcq_api.hip) || echo "base build failed"
  echo "built 00base ($BASE_REV)"
fi
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  OUT=$OUT/libmcq_$name.so $SRC/build.sh $defs
  echo "built $name ($defs)"
done
ls -la $OUT
