#!/bin/bash
# Builds libmcq.so (hand-written HIP kernels + C ABI) for gfx950 in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (no v_accvgpr_read before every LDS store of an updated tile)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -o libmcq.so mcq_kernels.hip mcq_api.hip "$@"
