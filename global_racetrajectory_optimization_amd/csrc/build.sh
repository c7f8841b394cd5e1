#!/bin/bash
# Builds libmcq.so (hand-written HIP kernels + C ABI) for gfx950 in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o libmcq.so mcq_kernels.hip mcq_api.hip "$@"
