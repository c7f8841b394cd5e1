#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "velocity_profile or lap_time" 2>&1 | grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP ver\|^ROCm" | tail -5
