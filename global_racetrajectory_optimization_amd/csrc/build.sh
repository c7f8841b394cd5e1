#!/bin/bash
# Builds libmcq.so (hand-written HIP kernels + C ABI) for gfx950 in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -enable-ipra=0: the AMDGPU backend's interprocedural register allocation is OFF.  With it, hipcc 7.2 miscompiled the callers of a
#   non-inlined device function once that function grew past the caller-saved VGPRs (the curvature-row path then read a null pointer on
#   the GPU -- found by the -m gpu suite, not by the SIMT emulator, which never sees the allocator).  Plain calling-convention clobbers
#   are also 1 % faster on the solver kernel.
# --gpu-max-threads-per-block=512: every device function of mcq_kernels.hip stays within 256 VGPRs (two workgroups of the solver kernel
#   per CU).
# ASM_OUT=<file>: also keep the device ISA of mcq_kernels.hip there (scripts/check_csr.py reads it instead of compiling a second time).
# -amdgpu-schedule-metric-bias=0: the machine scheduler weighs latency only, not occupancy (default bias 10) -- every function of this
#   library runs at the occupancy its register budget fixes (two workgroups per CU) and most of its time in dependent chains: +1.1 % on
#   the bench, same results bit for bit (max-ilp as the strategy: the tridiagonal sweeps 11 % faster, the elimination 2 % slower, +0.3 %).
# -O2, not -O3 (round 5): results bit for bit, factorisations 2.3 % faster, bench +1.6 % over four same-box pairs (101.2 / 99.4 / 101.3 / 100.8 k
#   against 98.7 / 98.4 / 100.8 / 98.2 k) -- -O3's extra unrolling of the elimination's and the chains' loops costs more scratch moves than it saves
#   (the tridiagonal sweeps lose 4 %: gradients 0.291 -> 0.303 ms per problem; they are a tenth of the factorisations).
FLAGS="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -mllvm -enable-ipra=${IPRA:-0} -mllvm -amdgpu-schedule-metric-bias=0"
OUT=${OUT:-libmcq.so}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
$HIPCC $FLAGS --gpu-max-threads-per-block=512 "$@" -c -o $TMP/kernels.o mcq_kernels.hip &
P1=$!
$HIPCC $FLAGS "$@" -c -o $TMP/api.o mcq_api.hip &
P2=$!
P3=
if [ -n "$ASM_OUT" ]; then
  $HIPCC $FLAGS --gpu-max-threads-per-block=512 "$@" -S --cuda-device-only -o "$ASM_OUT" mcq_kernels.hip &
  P3=$!
fi
wait $P1 || { echo "build.sh: mcq_kernels.hip failed to compile" >&2; exit 1; }
wait $P2 || { echo "build.sh: mcq_api.hip failed to compile" >&2; exit 1; }
if [ -n "$P3" ]; then wait $P3 || { echo "build.sh: ISA dump failed" >&2; exit 1; }; fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o $OUT $TMP/kernels.o $TMP/api.o -ldl      # (dlopen / dlsym of RCCL: glibc < 2.34 keeps them in libdl)
