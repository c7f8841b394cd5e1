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
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -mllvm -enable-ipra=0"
mkdir -p $OUT /tmp/mcq_base/csrc /tmp/mcq_base/include_dir
rm -f $OUT/*.so
BASE_REV=${BASE_REV:-436904a}
if [ "$BASE_REV" != "none" ]; then
  git -C $R show $BASE_REV:global_racetrajectory_optimization_amd/csrc/mcq_kernels.hip > /tmp/mcq_base/csrc/mcq_kernels.hip
  # kernels added to the C ABI since BASE_REV (not on the measured path)
  python3 - <<PY
import re
new = open("$SRC/mcq_kernels.hip").read()
k = new[new.index("// ---- fp32 rows, increment layout"):]
open("/tmp/mcq_base/csrc/mcq_kernels.hip", "a").write("\n" + k)
PY
  cp $SRC/mcq_kernels.h $SRC/mcq_api.hip /tmp/mcq_base/csrc/
  mkdir -p /tmp/mcq_base/include && cp $R/include/mcq.h /tmp/mcq_base/include/
  sed -i 's#"../../include/mcq.h"#"../include/mcq.h"#' /tmp/mcq_base/csrc/mcq_kernels.h
  (cd /tmp/mcq_base/csrc && $HIPCC $FLAGS -o $OUT/libmcq_00base.so mcq_kernels.hip mcq_api.hip)
  echo "built 00base ($BASE_REV kernels)"
fi
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  (cd $SRC && $HIPCC $FLAGS $defs -o $OUT/libmcq_$name.so mcq_kernels.hip mcq_api.hip)
  echo "built $name ($defs)"
done
ls -la $OUT
