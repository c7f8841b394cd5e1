#!/usr/bin/env python
"""Diagnostic: do the copies of mcq_solve_host_pipelined overlap the kernels?  Run under
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python scripts/diag_pipeline_trace.py
and look at the start / end timestamps of the copies against those of mcq_solve_kernel (scripts/diag_pipeline_trace.py --summarise <dir>)."""
import csv
import glob
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarise(d):
    ker = [r for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
    cop = [r for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
    sol = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in ker if r["Kernel_Name"].startswith("mcq_solve_kernel"))
    t0 = sol[0][0]
    print("solve kernels (ms from the first):", [(round((a - t0) / 1e6, 2), round((b - t0) / 1e6, 2)) for a, b in sol[-6:]])
    big = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "?"))) for r in cop
                 if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000)
    print("copies > 0.2 ms (start, end, direction):", [(round((a - t0) / 1e6, 2), round((b - t0) / 1e6, 2), n) for a, b, n in big[-12:]])
    inside = sum(1 for a, b, _ in big if any(s <= a and b <= e for s, e in sol))
    print("%d of %d large copies lie entirely inside a solve kernel's interval" % (inside, len(big)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        return summarise(sys.argv[2])
    from global_racetrajectory_optimization_amd import engine, synthetic
    B, n, steps = 1024, 2000, 6
    ref, nv, sc = synthetic.oval_batch(B, n=n)
    eng = engine.Engine(0)
    p_ref, p_nv, p_sc = eng.host_array((B, n, 4)), eng.host_array((B, n, 2)), eng.host_array((B, n))
    p_al = [eng.host_array((B, n)), eng.host_array((B, n))]
    p_ref[...], p_nv[...], p_sc[...] = ref, nv, sc
    eng.solve_host_pipelined([p_ref] * 2, [p_nv] * 2, [p_sc] * 2, 0.12, 3.4, p_al)
    t = time.perf_counter()
    eng.solve_host_pipelined([p_ref] * steps, [p_nv] * steps, [p_sc] * steps, 0.12, 3.4, [p_al[k & 1] for k in range(steps)])
    print("pipelined: %.2f ms per step" % (1e3 * (time.perf_counter() - t) / steps))
    eng.close()


if __name__ == "__main__":
    main()
