#!/bin/bash
# After scripts/gpu_r6_final.sh <tag> has run on the GPU box: copies the evidence the docs cite from gpurun_out/ (scratch) into profiles/
# (tracked) and regenerates the counter summary (profiles/<tag>_pmc.{md,json} and profiles/latest_pmc.json, stamped with the sources' SHA).
#   scripts/collect_profiles.sh <tag> "<title of the counter summary>"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=${1:-r06}
cd $R
# gpurun MERGES into gpurun_out/: counter / stats files of earlier runs of the same tag stay next to the new ones -- drop everything older
# than seven minutes before the newest file (a run of the counter passes takes four), so that the summary is of ONE run
python - "$T" <<'PY'
import glob, os, sys
dirs = [d for d in glob.glob("gpurun_out/%s_*" % sys.argv[1]) if os.path.isdir(d)]
files = [f for d in dirs for f in glob.glob(d + "/**/*", recursive=True) if os.path.isfile(f)]
if files:
    newest = max(os.path.getmtime(f) for f in files)
    for f in files:
        if os.path.getmtime(f) < newest - 420:
            os.remove(f)
PY
for f in bench.json bench_config4_1gpu.json bench_force_collective_1gpu.json config5_shard_1gpu.json shortest_path.json harness_replay.log \
         pytest_gpu.log pytest_gpu_poison.log kkt_check_n333_reference.txt write_bw.jsonl gi_and_comm_tests.txt pipeline_streams.txt gi_mode.json stress_two_paths.json stress_large_rings.json stress_large_rings_48GB.json ckpt_ablation.txt; do
  [ -f gpurun_out/${T}_$f ] && cp gpurun_out/${T}_$f profiles/${T}_$f
done
for k in kc_fused0 kc_fused1 kc_f32_fused0 kc_f32_fused1; do [ -f gpurun_out/${T}_$k.txt ] && cp gpurun_out/${T}_$k.txt profiles/${T}_kkt_check_$k.txt; done
# kernel stats: the newest *_kernel_stats.csv under the stats run
ks=$(ls -t gpurun_out/${T}_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$ks" ] && cp "$ks" profiles/${T}_kernel_stats.csv
python scripts/pmc_summary.py gpurun_out $T profiles/${T}_pmc "${2:-round 6: bench.py --no-extras (batch 1024, N = 2000), 1 x MI355X}"
ls -la profiles | grep ${T}_
