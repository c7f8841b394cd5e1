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
# mcq_kernels.hip is compiled TWICE (see its header): the library's kernels with the saddle-point core for two workgroups per CU
# (--gpu-max-threads-per-block=512: every device function stays within 256 VGPRs), and -DMCQ_CORE_BAND: namespace mcq_band, the solver
# kernel on the bordered-band core (shortest-path objective).
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -mllvm -enable-ipra=0"
OUT=${OUT:-libmcq.so}
TMP=$(mktemp -d)
$HIPCC $FLAGS --gpu-max-threads-per-block=512 "$@" -c -o $TMP/kkt.o mcq_kernels.hip &
$HIPCC $FLAGS -DMCQ_CORE_BAND "$@" -c -o $TMP/band.o mcq_kernels.hip &
$HIPCC $FLAGS "$@" -c -o $TMP/api.o mcq_api.hip &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $TMP/kkt.o $TMP/band.o $TMP/api.o
rm -rf $TMP
