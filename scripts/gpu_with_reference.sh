#!/bin/bash
# Runs a gpurun command with a SCRATCH copy of the reference tree shipped along (git-ignored scratch_ft/reference: python sources,
# inputs and params only), so that the `-m gpu` harness tests can drive the untouched main_globaltraj.py on the real library
# (BASELINE config 1).  The copy is deleted again as soon as the call returns; it is never committed.
#   scripts/gpu_with_reference.sh <gpurun-timeout-s> '<command run on the GPU box>'
set -u
mkdir -p scratch_ft gpurun_out
rm -rf scratch_ft/reference
mkdir -p scratch_ft/reference
(cd /root/reference && tar cf - --exclude=.git --exclude=outputs --exclude='*.png' --exclude='*.pdf' .) | (cd scratch_ft/reference && tar xf -)
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
rc=$?
rm -rf scratch_ft/reference
exit $rc
