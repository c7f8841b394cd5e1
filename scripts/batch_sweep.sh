#!/bin/bash
# GPU box: throughput of the headline workload against the batch size (launch tail / latency of a lone problem): bench.py --no-extras lines
for b in 1 64 256 512 1024 2048 4096; do
  python bench.py --no-extras --batch $b --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch', $b, 'solves/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],3), 'solver kernel ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3))"
done
