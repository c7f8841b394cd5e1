"""DIAGNOSTIC (round 5): do two launches of the solver kernel from two streams overlap their tails?  Two engines (= two handles, streams, workspaces) in one
process, the bench workload on each, launches enqueued alternately; against one engine doing the same number of launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from global_racetrajectory_optimization_amd import engine, synthetic
B, n, K = 1024, 2000, 10
ref, nv, sc = synthetic.oval_batch(B, n=n)
engs = [engine.Engine(0) for _ in range(2)]
bufs = []
for e in engs:
    d = dict(ref=e.alloc(ref.nbytes), nv=e.alloc(nv.nbytes), sc=e.alloc(sc.nbytes), al=e.alloc(8 * B * n), cu=e.alloc(8 * B), st=e.alloc(4 * B))
    e.upload(d["ref"], ref); e.upload(d["nv"], nv); e.upload(d["sc"], sc)
    bufs.append(d)
def run(which, steps):
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        i = which[k % len(which)]
        d = bufs[i]
        engs[i].solve_device(B, n, d["ref"], d["nv"], d["sc"], 0.12, 3.4, d["al"], d["cu"], d["st"])
    for e in engs: e.sync()
    return (time.perf_counter() - t0) / steps * 1e3
run([0, 1], 4)
print("one stream : %.3f ms per launch" % run([0], 2 * K))
print("two streams: %.3f ms per launch" % run([0, 1], 2 * K))
print("one stream : %.3f ms per launch" % run([0], 2 * K))
print("two streams: %.3f ms per launch" % run([0, 1], 2 * K))
for v in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "AMD_SERIALIZE_KERNEL"):
    print(v, os.environ.get(v))
