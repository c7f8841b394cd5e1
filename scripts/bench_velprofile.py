#!/usr/bin/env python
"""Measurement for row f-3: ggv velocity profile + lap time of 16384 (track, vehicle) variants -- BASELINE config 4's sweep
size -- on the racelines of the committed reference tracks; the host chain timed on a sample beside it.  One JSON line.

  python scripts/bench_velprofile.py [--variants 16384] [--cpu-sample 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from global_racetrajectory_optimization_amd import engine                                                              # noqa: E402
from global_racetrajectory_optimization_amd.trajectory_planning_helpers import (calc_ax_profile as ca, calc_head_curv_an as ch,  # noqa: E402
                                                                                calc_t_profile as ct, calc_vel_profile as cv,
                                                                                create_raceline as cr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", type=int, default=16384)
    ap.add_argument("--cpu-sample", type=int, default=16)
    args = ap.parse_args()
    eng = engine.Engine(0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "berlin_2018.npz"))
    out = cr.create_raceline(refline=g["reftrack"][:, :2], normvectors=g["normvec"], alpha=g["alpha"], stepsize_interp=3.0)
    _, kappa = ch.calc_head_curv_an(coeffs_x=out[2], coeffs_y=out[3], ind_spls=out[4], t_spls=out[5])
    el = out[8]
    n = kappa.size
    v = np.arange(0.0, 72.1, 4.0)
    ggv0 = np.column_stack((v, np.full(v.size, 12.0), np.full(v.size, 12.0)))
    axm = np.column_stack((v, np.interp(v, [0.0, 20.0, 72.0], [5.3, 5.3, 1.2])))
    bsz = args.variants
    scales = 0.3 + 0.7 * (np.arange(bsz) % 128) / 127.0
    tops = 100.0 / 3.6 + (50.0 / 3.6) * ((np.arange(bsz) // 128) % 128) / 127.0
    ggv = np.repeat(ggv0[None], bsz, axis=0)
    ggv[:, :, 1:] *= scales[:, None, None]
    axms = np.repeat(axm[None], bsz, axis=0)
    tr = np.zeros(bsz, dtype=np.int32)
    eng.vel_profile_batch(kappa[None], el[None], ggv[:64], axms[:64], 0.75, 1200.0, tops[:64], 1.0, tr[:64])     # warm-up
    t0 = time.perf_counter()
    vx, lt = eng.vel_profile_batch(kappa[None], el[None], ggv, axms, 0.75, 1200.0, tops, 1.0, tr)
    t_gpu = time.perf_counter() - t0                    # includes the PCIe copies of the tables and of the profiles
    ks = np.linspace(0, bsz - 1, args.cpu_sample).astype(int)
    t0 = time.perf_counter()
    worst, worst_t_host = 0.0, 0.0
    for k in ks:
        vx_h = cv.calc_vel_profile(ggv=ggv[k], ax_max_machines=axm, v_max=tops[k], kappa=kappa, el_lengths=el, closed=True,
                                   filt_window=None, dyn_model_exp=1.0, drag_coeff=0.75, m_veh=1200.0)
        ax_h = ca.calc_ax_profile(vx_profile=np.append(vx_h, vx_h[0]), el_lengths=el, eq_length_output=False)
        t_h = ct.calc_t_profile(vx_profile=vx_h, ax_profile=ax_h, el_lengths=el)
        vx_cl = np.append(vx_h, vx_h[0])
        worst = max(worst, float(np.max(np.abs(vx[k] - vx_h))), abs(float(lt[k] - np.sum(2.0 * el / (vx_cl[:-1] + vx_cl[1:])))))
        worst_t_host = max(worst_t_host, abs(float(lt[k] - t_h[-1])))
    t_cpu = (time.perf_counter() - t0) / len(ks)
    print(json.dumps({"variants": bsz, "n": int(n), "track": "berlin_2018 raceline", "gpu_seconds_incl_pcie": t_gpu,
                      "variants_per_s_gpu": bsz / t_gpu, "cpu_seconds_per_variant": t_cpu, "variants_per_s_cpu_1core": 1.0 / t_cpu,
                      "cpu_sample": len(ks), "max_abs_diff_vs_host": worst, "max_lap_time_diff_vs_host_unstable_formula_s": worst_t_host,
                      "lap_time_range_s": [float(lt.min()), float(lt.max())]}))


if __name__ == "__main__":
    main()
