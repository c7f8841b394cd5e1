#!/bin/bash
# Builds libmcq.so (hand-written HIP kernels + C ABI) for gfx950 in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (no v_accvgpr_read before every LDS store of an updated tile)
# -enable-ipra=0: the AMDGPU backend's interprocedural register allocation is OFF.  With it, hipcc 7.2 miscompiled the callers of a
#   non-inlined device function once that function grew past the caller-saved VGPRs (band_matvec with 16-byte loads: the curvature-row
#   path then read a null pointer on the GPU -- found by the -m gpu suite, not by the SIMT emulator, which never sees the allocator).
#   Plain calling-convention clobbers are also 1 % faster on the solver kernel.
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -mllvm -enable-ipra=0 -o libmcq.so mcq_kernels.hip mcq_api.hip "$@"
