"""bench.py's launch / sharding / collective logic without hardware (`--gpus N` must really start N ranks):
`python bench.py --gpus 2` with no launcher environment re-launches itself as two ranks under torch.distributed.run; here with the
SIMT-interpreted kernel library standing in for the GPU (bench.py --emulate: test only, the line says so) and -- round 5 -- the ENGINE'S OWN
collective path (mcq_comm_init / mcq_comm_allgather / mcq_comm_wait, the alternating send buffers, the lag-1 waits) on a shared-memory
stand-in for librccl (tests/stub/rccl_stub.cpp, $MCQ_RCCL_LIB); gloo is the rendezvous only, as on the GPU box.  The driver's own launcher
form (WORLD_SIZE set) is covered too."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(stub):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MCQ_LIB"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "2"
    env["MCQ_RCCL_LIB"] = stub       # the engine's collective (mcq_comm_*) on a shared-memory stand-in for RCCL: tests/stub/rccl_stub.cpp
    return env


def _check_line(stdout, n_gpus, steps=1):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout                       # exactly ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n_gpus and rec["config"]["ranks_seen"] == n_gpus
    assert rec["scaling"] == "weak" and rec["unit"] == "solves/s" and rec["steps"] == steps
    assert rec["config"]["failed_problems"] == 0
    assert rec["config"]["collective"].startswith("1 all-gather of alpha per step: ncclAllGather (RCCL) through the C ABI")
    assert "librccl_stub_sync.so" in rec["config"]["collective"]
    assert "EMULATED" in rec["data"]                     # never mistaken for a measurement
    assert rec["value"] > 0 and rec["config"]["rank_ms_per_step"]["max"] >= rec["config"]["rank_ms_per_step"]["min"] > 0
    return rec


def test_bench_gpus_2_self_launch(emu_lib, rccl_stub):
    # three timed steps behind one warm-up step: the two send buffers alternate, and mcq_comm_wait(h, 1) guards every reuse
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--n", "120",
           "--emulate", emu_lib, "--no-extras"]
    res = subprocess.run(cmd, env=_env(rccl_stub), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    _check_line(res.stdout, 2, steps=3)


def test_bench_under_the_drivers_launcher(emu_lib, rccl_stub):
    """The form the driver uses for N > 1: torch.distributed.run starts the ranks, bench.py reads RANK / WORLD_SIZE."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1",
           "--emulate", emu_lib, "--no-extras", "--config", "3"]
    env = _env(rccl_stub)
    env["MCQ_BENCH_TEST_N"] = "120"
    # (no --n here: without a "--" separator the launcher's argparse reports it as an ambiguous prefix of its own options; the
    # driver never passes it.  The ring size of the emulated run comes from the environment instead.)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    _check_line(res.stdout, 2)


def test_bench_config4_four_rank_shard(emu_lib, rccl_stub):
    """BASELINE config 4's 4-GPU shard (never run anywhere): `bench.py --config 4 --gpus 4` as four gloo ranks on the
    emulated library -- block partition of the (track x vehicle width) QPs, racelines and velocity profiles per rank, ONE all-gather
    of the lap times.  The matrix is shrunk to 2 tracks x 2 widths x 4 vehicles so that the interpreter finishes: one QP per rank."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "4", "--steps", "1", "--warmup", "0",
           "--c4-tracks", "rounded_rectangle,handling_track", "--c4-widths", "2", "--c4-vehicles", "4", "--emulate", emu_lib, "--no-extras"]
    res = subprocess.run(cmd, env=_env(rccl_stub), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 4 and rec["config"]["ranks_seen"] == 4 and rec["scaling"] == "strong" and rec["unit"] == "variants/s"
    assert rec["config"]["variants_total"] == 16 and rec["config"]["variants_this_rank"] == 4
    assert "EMULATED" in rec["data"]
    # every rank's lap times arrived, in partition order, and are the single-process ones
    laps = np.array(rec["config"]["lap_times_gathered_s"])
    assert laps.shape == (16,) and np.all(np.isfinite(laps)) and np.all(laps > 5.0)
    cmd1 = [c for c in cmd]
    cmd1[cmd1.index("--gpus") + 1] = "1"
    res1 = subprocess.run(cmd1, env=_env(rccl_stub), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res1.returncode == 0, res1.stderr[-2000:]
    rec1 = json.loads([l for l in res1.stdout.splitlines() if l.strip()][0])
    assert np.array_equal(np.array(rec1["config"]["lap_times_gathered_s"]), laps)


def test_bench_config5_f32_two_rank(emu_lib, rccl_stub):
    """BASELINE config 5's path on two gloo ranks: per-track centrelines, float increment rows + fp64 origins in "HBM", float alpha,
    ONE all-gather of the float alpha (half the bytes of the fp64 collective)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "5", "--steps", "1", "--warmup", "0", "--batch", "1",
           "--n", "120", "--emulate", emu_lib, "--no-extras"]
    res = subprocess.run(cmd, env=_env(rccl_stub), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    rec = _check_line(res.stdout, 2)
    assert rec["config"]["io"].startswith("f32 rows (ring increments + fp64 origin)")
    assert rec["config"]["centrelines"] == "perturbed per track" and rec["config"]["allgather_dtype"] == "float32"
