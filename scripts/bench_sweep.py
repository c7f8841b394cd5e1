#!/usr/bin/env python
"""BASELINE config 4 end to end on one GPU: the lap-time matrix of 16384 variants = 4 reference tracks (berlin_2018 N = 776,
modena_2019 N = 663, handling_track N = 208, rounded_rectangle N = 105 -- the reference ships no Monza) x 64 vehicle widths
(w_veh 2.0 ... 3.4 m) x 64 (gg-scale, top-speed) vehicles.  Per variant the reference would run opt_min_curv -> create_raceline
-> calc_head_curv_an -> calc_vel_profile -> calc_ax_profile / calc_t_profile [REF main_globaltraj.py:264-271, 371-422].

The vehicle tables do not enter the QP (only w_veh moves the box), so two ways of filling the matrix are measured and must agree:

  naive   every variant solves its own QP: 16384 ragged QPs in one mcq_solve_batch launch
  shared  256 unique (track, w_veh) QPs, racelines + curvature of the 256 on the device (mcq_raceline_device), 16384 velocity
          profiles over them in one ragged launch (mcq_vel_profile_device_ragged)

(SURVEY.md section 8d's "shared-factor" reuse does not carry over: the solver here is interior point + exact active set, whose
factorisations depend on the box; what IS shared per (track, w_veh) is the whole QP.)

Checks: naive alpha == shared alpha bitwise (one workgroup per problem, no cross-problem arithmetic); a sample of the QPs
against the live dense oracle; a sample of the lap times against the host chain.  One JSON line.

  python scripts/bench_sweep.py [--widths 64] [--vehicles 64] [--oracle-sample 6] [--skip-naive]
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
from global_racetrajectory_optimization_amd.trajectory_planning_helpers import (calc_head_curv_an as ch,              # noqa: E402
                                                                                calc_vel_profile as cv, create_raceline as cr)

TRACKS = ("berlin_2018", "modena_2019", "handling_track", "rounded_rectangle")
KAPPA_BOUND, STEP_OUT = 0.12, 2.0          # curvlim [REF params/racecar.ini:49], stepsize_interp_after_opt [REF params/racecar.ini:15]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--widths", type=int, default=64)
    ap.add_argument("--vehicles", type=int, default=64)
    ap.add_argument("--oracle-sample", type=int, default=6)
    ap.add_argument("--skip-naive", action="store_true")
    ap.add_argument("--tracks", default=",".join(TRACKS))
    args = ap.parse_args()
    tracks = tuple(args.tracks.split(","))
    eng = engine.Engine(0)
    gold = [np.load(os.path.join(ROOT, "tests", "golden", t + ".npz")) for t in tracks]
    w_grid = np.linspace(2.0, 3.4, args.widths)
    nveh = args.vehicles
    v = np.arange(0.0, 72.1, 4.0)
    ggv0 = np.column_stack((v, np.full(v.size, 12.0), np.full(v.size, 12.0)))
    axm0 = np.column_stack((v, np.interp(v, [0.0, 20.0, 72.0], [5.3, 5.3, 1.2])))
    side = int(round(np.sqrt(nveh)))
    gg_scale = 0.3 + 0.7 * (np.arange(nveh) % side) / max(side - 1, 1)
    v_top = 100.0 / 3.6 + (150.0 / 3.6) * (np.arange(nveh) // side) / max((nveh - 1) // side, 1)

    # ---- the 256 unique QPs ------------------------------------------------------------------------------------------------
    uniq = []
    for ti, g in enumerate(gold):
        for w in w_grid:
            uniq.append(dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=KAPPA_BOUND,
                             w_veh=float(w), track=ti))
    eng.solve_batch(uniq[:4])                                                   # warm-up (workspace, code objects)
    t0 = time.perf_counter()
    al_u, curv_u, st_u, info_u = eng.solve_batch(uniq)
    t_qp_shared = time.perf_counter() - t0
    assert np.all(st_u == 0), "unique QPs: status %s" % np.unique(st_u)

    # ---- racelines + curvature of the 256 on the device, then 16384 velocity profiles over them --------------------------------
    t0 = time.perf_counter()
    race = eng.raceline_batch([p["reftrack"] for p in uniq], [p["normvec"] for p in uniq], al_u, STEP_OUT)
    t_race = time.perf_counter() - t0
    assert np.all(race["status"] == 0)
    nvar = len(uniq) * nveh
    track_of = np.repeat(np.arange(len(uniq), dtype=np.int32), nveh)
    veh_of = np.tile(np.arange(nveh), len(uniq))
    ggv = np.repeat(ggv0[None], nvar, axis=0)
    ggv[:, :, 1:] *= gg_scale[veh_of][:, None, None]
    axm = np.repeat(axm0[None], nvar, axis=0)
    tops = v_top[veh_of]
    t0 = time.perf_counter()
    vx, lap = eng.vel_profile_batch(race["kappa"], race["el_lengths"], ggv, axm, 0.75, 1200.0, tops, 1.0, track_of=track_of,
                                    n_of_track=race["m"])
    t_vel = time.perf_counter() - t0
    assert np.all(np.isfinite(lap))

    # ---- naive: every variant its own QP --------------------------------------------------------------------------------------
    t_qp_naive, naive_equal, mean_n = None, None, float(np.mean([p["reftrack"].shape[0] for p in uniq]))
    if not args.skip_naive:
        probs = [uniq[k] for k in track_of]
        t0 = time.perf_counter()
        al_n, _, st_n, _ = eng.solve_batch(probs)
        t_qp_naive = time.perf_counter() - t0
        assert np.all(st_n == 0)
        naive_equal = all(np.array_equal(al_n[j], al_u[track_of[j]]) for j in range(nvar))

    # ---- checks against the CPU chain on a sample (test infrastructure: oracle/) ----------------------------------------------------
    from oracle import tph_ref
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs
    rng = np.random.default_rng(4)
    worst_alpha, t_oracle = 0.0, 0.0
    pick = [int(k) for k in rng.choice(len(uniq), size=min(args.oracle_sample, len(uniq)), replace=False)]
    for k in pick:
        p = uniq[k]
        A = cs.build_les_matrix(p["reftrack"].shape[0], p["scaling"])
        t0 = time.perf_counter()
        a_ref, _ = tph_ref.opt_min_curv(p["reftrack"], p["normvec"], A, KAPPA_BOUND, p["w_veh"])
        t_oracle += time.perf_counter() - t0
        worst_alpha = max(worst_alpha, float(np.max(np.abs(a_ref - al_u[k]))))
    worst_lap, worst_vx, t_host_chain = 0.0, 0.0, 0.0
    for j in [int(k) for k in rng.choice(nvar, size=8, replace=False)]:
        p = uniq[track_of[j]]
        t0 = time.perf_counter()
        out = cr.create_raceline(refline=p["reftrack"][:, :2], normvectors=p["normvec"], alpha=al_u[track_of[j]], stepsize_interp=STEP_OUT)
        _, kap = ch.calc_head_curv_an(coeffs_x=out[2], coeffs_y=out[3], ind_spls=out[4], t_spls=out[5])
        el = out[8]
        vx_h = cv.calc_vel_profile(ggv=ggv[j], ax_max_machines=axm0, v_max=tops[j], kappa=kap, el_lengths=el, closed=True,
                                   filt_window=None, dyn_model_exp=1.0, drag_coeff=0.75, m_veh=1200.0)
        t_host_chain += time.perf_counter() - t0
        vx_cl = np.append(vx_h, vx_h[0])
        worst_vx = max(worst_vx, float(np.max(np.abs(vx[j, :kap.size] - vx_h))))
        worst_lap = max(worst_lap, abs(float(lap[j] - np.sum(2.0 * el / (vx_cl[:-1] + vx_cl[1:])))))

    t_shared = t_qp_shared + t_race + t_vel
    print(json.dumps({
        "workload": "BASELINE config 4 on 1 GPU: %d tracks x %d vehicle widths x %d vehicles = %d variants" % (len(tracks), args.widths, nveh, nvar),
        "tracks": {t: int(g["reftrack"].shape[0]) for t, g in zip(tracks, gold)}, "mean_waypoints": mean_n,
        "shared": {"unique_qps": len(uniq), "qp_seconds": t_qp_shared, "raceline_seconds": t_race, "vel_profile_seconds": t_vel,
                   "total_seconds": t_shared, "variants_per_s": nvar / t_shared},
        "naive": None if t_qp_naive is None else {"qps": nvar, "qp_seconds": t_qp_naive, "qp_solves_per_s": nvar / t_qp_naive,
                                                  "total_seconds": t_qp_naive + t_race + t_vel,
                                                  "variants_per_s": nvar / (t_qp_naive + t_race + t_vel),
                                                  "alpha_bitwise_equal_to_shared": naive_equal},
        "timings_include": "host packing + PCIe of every host-buffer entry point",
        "mean_ipm_iters": float(np.mean([i["ipm_iters"] for i in info_u])), "mean_as_iters": float(np.mean([i["as_iters"] for i in info_u])),
        "lap_time_range_s": [float(lap.min()), float(lap.max())],
        "checks": {"oracle_qps": len(pick), "max_abs_alpha_diff_vs_dense_oracle_m": worst_alpha, "oracle_seconds_per_qp": t_oracle / max(len(pick), 1),
                   "host_chain_variants": 8, "max_abs_vx_diff_vs_host_chain": worst_vx, "max_abs_lap_time_diff_vs_host_chain_s": worst_lap,
                   "host_chain_seconds_per_variant_excl_qp": t_host_chain / 8}}))
    eng.close()


if __name__ == "__main__":
    main()
