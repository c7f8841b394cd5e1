#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 300 python scripts/diag_stadium.py 2>&1 | grep status | cut -c1-250
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
scripts/gpu_variants.sh r04d "n2000_first_pass"
