#!/bin/bash
# Builds the ablation variants of scripts/factor_bench.hip into build/fb/ (git-ignored, shipped by gpurun):
#   scripts/build_factor_bench.sh "<name> <extra -D flags>" ...      e.g.  "a512 -DMCQ_ABL=1536"
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/fb
rm -f $R/build/fb/fb_*
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -enable-ipra=0 -DMCQ_CORE_BAND "$@" -o $R/build/fb/fb_$name $R/scripts/factor_bench.hip 2>&1 | grep -i "error" &
done
wait
ls $R/build/fb
