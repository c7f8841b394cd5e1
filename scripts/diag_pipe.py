"""DIAGNOSTIC (round 5): mcq_solve_host_pipelined, 20 steps of the bench workload, against the device-resident loop on the same box.  MCQ_PIPE_ONE_STREAM=1: round 4's form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from global_racetrajectory_optimization_amd import engine, synthetic
B, n, K = 1024, 2000, 20
ref, nv, sc = synthetic.oval_batch(B, n=n)
eng = engine.Engine(0)
d = dict(ref=eng.alloc(ref.nbytes), nv=eng.alloc(nv.nbytes), sc=eng.alloc(sc.nbytes), al=eng.alloc(8 * B * n), cu=eng.alloc(8 * B), st=eng.alloc(4 * B))
eng.upload(d["ref"], ref); eng.upload(d["nv"], nv); eng.upload(d["sc"], sc)
for rep in range(2):
    eng.sync(); t0 = time.perf_counter()
    for k in range(K):
        eng.solve_device(B, n, d["ref"], d["nv"], d["sc"], 0.12, 3.4, d["al"], d["cu"], d["st"])
    eng.sync()
    print("device resident: %.3f ms per step" % ((time.perf_counter() - t0) / K * 1e3))
p_ref, p_nv, p_sc = eng.host_array((B, n, 4)), eng.host_array((B, n, 2)), eng.host_array((B, n))
p_al = [eng.host_array((B, n)), eng.host_array((B, n))]
p_ref[...], p_nv[...], p_sc[...] = ref, nv, sc
eng.solve_host_pipelined([p_ref] * 2, [p_nv] * 2, [p_sc] * 2, 0.12, 3.4, p_al)
for rep in range(3):
    t0 = time.perf_counter()
    eng.solve_host_pipelined([p_ref] * K, [p_nv] * K, [p_sc] * K, 0.12, 3.4, [p_al[k & 1] for k in range(K)])
    print("pipelined (MCQ_PIPE_ONE_STREAM=%s): %.3f ms per step" % (os.environ.get("MCQ_PIPE_ONE_STREAM"), (time.perf_counter() - t0) / K * 1e3))
