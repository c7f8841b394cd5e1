"""The engine's collective path with MORE THAN ONE RANK on the one GPU a test box has (until round 5 every ordering
of that path -- the comm stream behind the compute stream, the event ring, mcq_comm_wait(h, 1) guarding two alternating send buffers -- had only
ever run with world = 1, where no ordering bug can show).  Two processes share GPU 0; the fabric is tests/stub/librccl_stub.so, an ASYNCHRONOUS
shared-memory stand-in for the five RCCL entry points the engine binds (stream-ordered copies and host functions; $MCQ_RCCL_LIB).  Real RCCL
with more than one rank needs more than one GPU: that is the driver's SCALE run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub():
    d = os.path.join(ROOT, "tests", "stub")
    path = os.path.join(d, "librccl_stub.so")
    if not os.path.exists(path) or os.path.getmtime(os.path.join(d, "rccl_stub.cpp")) > os.path.getmtime(path):
        subprocess.run([os.path.join(d, "build_stub.sh")], check=True)
    assert os.path.exists(path), "tests/stub/build_stub.sh did not produce librccl_stub.so (hipcc?)"
    return path


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["MCQ_RCCL_LIB"] = _stub()
    return env


def test_bench_two_ranks_on_one_gpu_through_the_engines_collective():
    """bench.py --gpus 2 as the driver's SCALE run starts it, both ranks on GPU 0 (the one-visible-device fallback): four timed steps behind a
    warm-up step, the all-gather of step k on the comm stream while step k + 1 solves into the other buffer.  bench.py itself asserts that the
    gathered tensor holds this rank's shard bitwise and that the communicator is (rank, world); here: one JSON line, both ranks seen, no failed
    problem, a gather time measured by mcq_comm_wait."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "64", "--no-extras"]
    res = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["ranks_seen"] == 2 and rec["config"]["failed_problems"] == 0
    assert rec["config"]["collective"].startswith("1 all-gather of alpha per step: ncclAllGather (RCCL) through the C ABI")
    assert rec["config"]["allgather_ms"] is not None and rec["config"]["allgather_ms"] > 0.0
    print("two ranks on one GPU: %.0f solves/s aggregate, gather %.3f ms (shared-memory stand-in: not a fabric measurement)" % (
        rec["value"], rec["config"]["allgather_ms"]))


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
from global_racetrajectory_optimization_amd import engine, parallel, synthetic
rank, world = int(sys.argv[1]), 2
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
eng = engine.Engine(0)
assert parallel.init_engine_comm(eng, dist) == (rank, world)
# alternating send buffers, lag-1 waits: six gathers of 4 MB, every one checked after the fact
n = 1 << 19
d_s = [eng.alloc(8 * n) for _ in range(2)]
d_r = [eng.alloc(8 * n * world) for _ in range(6)]
for k in range(6):
    if k >= 2:
        eng.comm_wait(1)                      # the gather that last read this send buffer
    eng.upload(d_s[k % 2], np.full(n, 1000.0 * k + rank))
    eng.comm_allgather(d_s[k % 2], d_r[k], n, eng.DT_F64)
eng.comm_wait(0)
for k in range(6):
    got = eng.download(d_r[k], (world, n), np.float64)
    assert np.all(got[0] == 1000.0 * k) and np.all(got[1] == 1000.0 * k + 1), (rank, k, got[:, :2])
# the sharded solve: 5 ragged problems over 2 ranks, every rank ends with the full batch
probs = [dict(reftrack=synthetic.oval_batch(1, n=400 - 7 * k, first=40 + k, perturb_centreline=True)[0][0], normvec=None, scaling=None,
              kappa_bound=0.5, w_veh=2.0 + 0.2 * k) for k in range(5)]
a, c, s = parallel.solve_sharded(probs, eng, dist=dist)
np.savez(sys.argv[3], status=s, curv=c, **{{"a%d" % k: a[k] for k in range(5)}})
dist.barrier()
eng.close()
'''


def test_two_ranks_sharded_solve_and_lagged_gathers(gpu_engine, tmp_path):
    import socket
    from global_racetrajectory_optimization_amd import synthetic
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    outs = [str(tmp_path / ("out%d.npz" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(port), outs[r]], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    probs = [dict(reftrack=synthetic.oval_batch(1, n=400 - 7 * k, first=40 + k, perturb_centreline=True)[0][0], normvec=None, scaling=None,
                  kappa_bound=0.5, w_veh=2.0 + 0.2 * k) for k in range(5)]
    a1, c1, s1, _ = gpu_engine.solve_batch(probs)
    for r in range(2):
        z = np.load(outs[r])
        assert list(z["status"]) == list(s1) == [0] * 5 and np.array_equal(z["curv"], c1)
        for k in range(5):
            assert np.array_equal(z["a%d" % k], a1[k]), (r, k)


def test_pipelined_host_entry_on_two_compute_streams_is_bitwise_the_blocking_entry(gpu_engine):
    """mcq_solve_host_pipelined (round 5: consecutive steps on two compute streams with a workspace each, curv / status through pinned staging):
    seven steps of DIFFERENT tracks -- pinned and pageable buffers mixed, normals given / derived -- against mcq_solve_host step by step, bitwise;
    one step holds a problem the block-pivoting phase is not allowed to finish (max_as_iter = 1), so that the second workspace's
    Goldfarb-Idnani slots are used as well."""
    from global_racetrajectory_optimization_amd import synthetic
    eng = gpu_engine
    B, n, K = 48, 600, 7
    refs, nvs, scs, outs = [], [], [], []
    for k in range(K):
        ref, nv, sc = synthetic.oval_batch(B, n=n, first=300 + 50 * k, perturb_centreline=True)
        if k % 2 == 0:          # pinned
            pr, pn, ps = eng.host_array((B, n, 4)), eng.host_array((B, n, 2)), eng.host_array((B, n))
            pr[...], pn[...], ps[...] = ref, nv, sc
            refs.append(pr); nvs.append(pn if k % 4 == 0 else None); scs.append(ps if k % 4 == 0 else None)
        else:                   # pageable
            refs.append(ref); nvs.append(nv); scs.append(sc)
        outs.append(eng.host_array((B, n)) if k % 3 else np.empty((B, n)))
    for opt in (dict(), dict(max_as_iter=1)):
        for o in outs:
            o[...] = np.nan
        curv, st = eng.solve_host_pipelined(refs, nvs, scs, 0.12, 3.4, outs, **opt)
        assert np.all(st == 0), np.unique(st)
        ran = 0
        for k in range(K):
            a1, c1, s1, inf = eng.solve_host(refs[k], nvs[k], scs[k], 0.12, 3.4, **opt)
            assert np.array_equal(a1, outs[k]) and np.array_equal(c1, curv[k]) and np.array_equal(s1, st[k]), (opt, k)
            ran += sum(1 for i in inf if i.gi_iters > 0)
        if opt:
            assert ran > 0, "max_as_iter = 1 sent no problem through the Goldfarb-Idnani path: the test exercises nothing"


def test_resident_stream_entry_on_two_compute_streams_is_bitwise_the_single_launches(gpu_engine):
    """mcq_solve_device_stream: five resident batches of different tracks through one call that alternates the engine's two compute streams,
    against mcq_solve_device launch by launch -- bitwise; the call is asynchronous and joins the second stream back into the first (a download
    right behind it sees every step)."""
    from global_racetrajectory_optimization_amd import synthetic
    eng = gpu_engine
    B, n, K = 64, 700, 5
    d_ref, d_nv, d_sc, d_al, d_cu, d_st, want = [], [], [], [], [], [], []
    for k in range(K):
        ref, nv, sc = synthetic.oval_batch(B, n=n, first=900 + 64 * k, perturb_centreline=True)
        p = [eng.alloc(a.nbytes) for a in (ref, nv, sc)]
        for q, a in zip(p, (ref, nv, sc)):
            eng.upload(q, a)
        d_ref.append(p[0]); d_nv.append(p[1] if k != 2 else None); d_sc.append(p[2] if k != 2 else None)
        d_al.append(eng.alloc(8 * B * n)); d_cu.append(eng.alloc(8 * B)); d_st.append(eng.alloc(4 * B))
        eng.solve_device(B, n, p[0], d_nv[-1], d_sc[-1], 0.12, 3.4, d_al[-1], d_cu[-1], d_st[-1])
        want.append((eng.download(d_al[-1], (B, n), np.float64), eng.download(d_cu[-1], (B,), np.float64), eng.download(d_st[-1], (B,), np.int32)))
        eng.upload(d_al[-1], np.full((B, n), np.nan))
    eng.solve_device_stream(B, n, d_ref, d_nv, d_sc, 0.12, 3.4, d_al, d_cu, d_st)
    for k in range(K):          # (no sync() in between: the blocking download is ordered behind the call)
        a = eng.download(d_al[k], (B, n), np.float64)
        assert np.array_equal(a, want[k][0]) and np.array_equal(eng.download(d_cu[k], (B,), np.float64), want[k][1]), k
        assert np.all(eng.download(d_st[k], (B,), np.int32) == 0)
    for p in d_ref + [q for q in d_nv if q] + [q for q in d_sc if q] + d_al + d_cu + d_st:
        eng.free(p)


def test_iqp_rounds_in_one_launch_are_bitwise_the_round_by_round_loop(gpu_engine, monkeypatch):
    """mcq_iqp_batch with the first iters_min rounds as ONE launch (mcq_iqp_rounds_kernel: a workgroup takes its track through the rounds on
    its own; the default) against the one-launch-per-round loop ($MCQ_IQP_FUSED=0): 192 ovals of 600 waypoints with per-track centrelines --
    some need a fourth round, which the loop runs afterwards --, one track narrower than the vehicle (it stops in round 1); the batch packed
    by 1 and by 5 host threads, as a list of tracks and as one dict of stacked arrays: end states, round counts, curvature errors bitwise."""
    from global_racetrajectory_optimization_amd import engine, synthetic
    eng = gpu_engine
    B, n = 192, 600
    ref, nv, sc = synthetic.oval_batch(B, n=n, first=4000, perturb_centreline=True)
    ref[77, :, 2:] = 1.0
    trk = [dict(reftrack=ref[k], normvectors=nv[k], scaling=sc[k]) for k in range(B)]
    res = {}
    for mode, fused, threads, form in (("loop", "0", "1", trk), ("one launch", "1", "5", trk),
                                       ("one launch, stacked", "1", "1", dict(reftrack=ref, normvectors=nv, scaling=sc))):
        monkeypatch.setenv("MCQ_IQP_FUSED", fused)
        monkeypatch.setenv("MCQ_PACK_THREADS", threads)
        # (curv_error_allowed 4.5e-5: on these gentle ovals the third pass leaves 4e-5 .. 5.5e-5, the fourth 2.5e-5 .. 3.3e-5)
        res[mode] = eng.iqp_batch(form, 0.12, 3.4, 3.0, iters_min=3, curv_error_allowed=4.5e-5, max_rounds=6)
    a = res["loop"]
    assert a["status"][77] == engine.STATUS_INFEASIBLE and a["rounds"][77] == 1
    assert np.count_nonzero(a["status"] == engine.STATUS_INFEASIBLE) == 1 and a["rounds"].max() > 3 and np.count_nonzero(a["rounds"] == 3) > 0, (
        np.unique(a["status"], return_counts=True), np.unique(a["rounds"], return_counts=True))
    for mode in ("one launch", "one launch, stacked"):
        b = res[mode]
        assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["rounds"], b["rounds"]) and np.array_equal(a["n"], b["n"])
        assert np.array_equal(a["curv_err"], b["curv_err"]) and np.array_equal(a["curv_trace"], b["curv_trace"])
        for k in range(B):
            assert np.array_equal(a["alpha"][k], b["alpha"][k]) and np.array_equal(a["reftrack"][k], b["reftrack"][k]), (mode, k)
            assert np.array_equal(a["normvectors"][k], b["normvectors"][k]), (mode, k)
        assert a["stats"]["qp_solves"] == b["stats"]["qp_solves"]
