#!/bin/bash
# Round 6 evidence call: the whole -m gpu suite (plain and with poisoned workspaces / LDS; the replay of main_globaltraj.py's recorded boundary
# calls included), the bench line, shortest path, force-collective, config 4 and config 5 shard on one GPU,
# the elimination in isolation, the counter passes (scripts/profile_round.sh).
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r06}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/${T}_pytest_gpu.log
tail -14 gpurun_out/${T}_pytest_gpu.log
MCQ_POISON=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_gi.py -m gpu -q > gpurun_out/${T}_pytest_gpu_poison.log 2>&1
echo "poison pytest rc $?" >> gpurun_out/${T}_pytest_gpu_poison.log
tail -3 gpurun_out/${T}_pytest_gpu_poison.log
# config 1 on the real library, with the script's own output: the untouched main_globaltraj.py (mincurv, mincurv_iqp), stamped with the
# SHA-256 of the engine sources it ran on
(echo "engine sources sha256: $(python -c 'import bench; print(bench.source_sha())')"; timeout 600 python -m pytest tests/test_harness.py -m gpu -q -s 2>&1) > gpurun_out/${T}_harness_replay.log
echo "harness rc $?"; grep -c "replay of main_globaltraj" gpurun_out/${T}_harness_replay.log
(timeout 300 python -m pytest tests/test_gpu_gi.py tests/test_gpu_comm.py -m gpu -q -s 2>&1 | grep -E "GI mode|stadium 360|curvature-tight|the same through|rings above|two ranks|passed|failed") > gpurun_out/${T}_gi_and_comm_tests.txt
(timeout 120 python scripts/diag_pipe.py; MCQ_PIPE_ONE_STREAM=1 timeout 120 python scripts/diag_pipe.py) > gpurun_out/${T}_pipeline_streams.txt 2>&1
timeout 300 python scripts/stress_two_paths.py 3000 11 > gpurun_out/${T}_stress_two_paths.json 2> gpurun_out/${T}_stress.err
echo "stress rc $?"
timeout 400 python scripts/stress_two_paths.py 600 23 1000 3000 > gpurun_out/${T}_stress_large_rings.json 2> gpurun_out/${T}_stress_large.err
echo "large rings rc $?"
MCQ_GI_BYTES=51539607552 timeout 400 python scripts/stress_two_paths.py 600 23 1000 3000 > gpurun_out/${T}_stress_large_rings_48GB.json 2> gpurun_out/${T}_stress_large48.err
echo "large rings, MCQ_GI_BYTES = 48 GB, rc $?"
timeout 300 python scripts/bench_gi_mode.py > gpurun_out/${T}_gi_mode.json 2> gpurun_out/${T}_gi_mode.err
echo "gi mode rc $?"
timeout 300 python scripts/bench_shortest_path.py > gpurun_out/${T}_shortest_path.json 2> gpurun_out/${T}_shortest_path.err
echo "shortest path rc $?"
timeout 600 python bench.py --config 4 --steps 3 --warmup 1 > gpurun_out/${T}_bench_config4_1gpu.json 2> gpurun_out/${T}_bench_config4.err
echo "config4 rc $?"; cut -c1-200 gpurun_out/${T}_bench_config4_1gpu.json
timeout 600 python bench.py --force-collective --steps 5 --warmup 2 --no-extras > gpurun_out/${T}_bench_force_collective_1gpu.json 2> gpurun_out/${T}_bench_fc.err
echo "force-collective rc $?"
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-extras > gpurun_out/${T}_config5_shard_1gpu.json 2> gpurun_out/${T}_config5.err
echo "config5 rc $?"; cut -c1-200 gpurun_out/${T}_config5_shard_1gpu.json
for k in kc kc_f32; do for fused in 1 0; do timeout 120 ./build/kc/$k 2000 6 2048 12 0 $fused > gpurun_out/${T}_${k}_fused${fused}.txt 2>&1; done; done
timeout 300 ./build/kc/kc 333 20 8 14 1 0 > gpurun_out/${T}_kkt_check_n333_reference.txt 2>&1
[ -x build/tools/write_bw ] && timeout 120 build/tools/write_bw 736 5 > gpurun_out/${T}_write_bw.jsonl 2>&1
for k in kc kc_abl_noaystore kc_abl_noayload kc_abl_noayboth; do for rep in 1 2; do echo "== $k fused=1 run $rep"; timeout 120 ./build/kc/$k 2000 6 2048 12 0 1 2>&1 | head -3; done; done > gpurun_out/${T}_ckpt_ablation.txt 2>&1
scripts/profile_round.sh $T 2>&1 | grep "pmc\|calib" | tr '\n' ' '
# the counter summary of THIS run next to the sources it ran on (profiles/latest_pmc.json carries their SHA-256), then the bench line: its
# roofline.traffic quotes that summary (scripts/collect_profiles.sh regenerates the same files from the merged gpurun_out/)
python scripts/pmc_summary.py gpurun_out $T profiles/${T}_pmc "round 6: bench.py --no-extras (batch 1024, N = 2000), 1 x MI355X" > /dev/null 2>&1
echo "pmc summary rc $?"
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc $?"; cut -c1-300 gpurun_out/${T}_bench.json
