#!/bin/bash
# The scaling table north_star asks for ("throughput reported at 1/2/4/8 GPUs as absolute solves/sec and as fraction of the roofline"), for the day
# an 8-GPU node is at hand (gpurun grants one GPU; the driver's SCALE run has found no node in rounds 1-6: NO number below has ever been produced
# on more than one GPU).  Run on the node, from the repo root:
#     scripts/scale_curve.sh [config: 3 (default) | 4 | 5] [steps 10] [warmup 2]
# For every N in 1 2 4 8 (as many as the node has): bench.py under torch.distributed.run exactly as the driver launches it, one JSON line per N into
# gpurun_out/scale_c<config>_n<N>.json, then the table: N, solves/s (whole job), per-GPU rate, efficiency against N = 1, roofline fraction of rank 0's
# solver kernel, the ranks' own ms per step (min / max), the all-gather's device time.
set -u
CFG=${1:-3}; STEPS=${2:-10}; WARM=${3:-2}
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python - <<'PY'
import ctypes, sys
try:
    hip = ctypes.CDLL("libamdhip64.so"); n = ctypes.c_int(0); hip.hipGetDeviceCount(ctypes.byref(n)); print(n.value)
except Exception:
    print(0)
PY
)
echo "devices visible: $NGPU"
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || continue
  OUT=gpurun_out/scale_c${CFG}_n${N}.json
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --config $CFG --steps $STEPS --warmup $WARM --no-extras > $OUT 2> ${OUT%.json}.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --gpus $N --config $CFG --steps $STEPS --warmup $WARM --no-extras > $OUT 2> ${OUT%.json}.err
  fi
  echo "N=$N rc $?"
done
python - "$CFG" <<'PY'
import json, os, sys
cfg = sys.argv[1]
rows, base = [], None
for n in (1, 2, 4, 8):
    p = "gpurun_out/scale_c%s_n%d.json" % (cfg, n)
    if not os.path.exists(p) or os.path.getsize(p) == 0:
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    c = d["config"]
    base = base or d["value"] / d["n_gpus"]
    rk = c.get("rank_ms_per_step", {})
    rows.append((d["n_gpus"], d["value"], d["value"] / d["n_gpus"], d["value"] / d["n_gpus"] / base, (d.get("roofline") or {}).get("frac"),
                 rk.get("min"), rk.get("max"), c.get("allgather_ms"), c.get("ranks_seen")))
print("| GPUs | %s (whole job) | per GPU | efficiency vs N=1 | roofline frac (rank 0) | rank ms/step min | max | all-gather ms | ranks seen |" % (rows and "value" or "-"))
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %d | %.0f | %.0f | %.3f | %s | %s | %s | %s | %s |" % (r[0], r[1], r[2], r[3], "%.3f" % r[4] if r[4] else "-", "%.3f" % r[5] if r[5] else "-",
                                                                 "%.3f" % r[6] if r[6] else "-", "%.3f" % r[7] if r[7] else "-", r[8]))
PY
