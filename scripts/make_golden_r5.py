"""
Round-5 additions to tests/golden/ (existing fixtures are NOT regenerated).  Everything here goes through oracle/tph_ref (dense 4N x 4N
inverse, dense products, all 4N rows) + the dense Goldfarb-Idnani of oracle/gi_dense.c.

  kappa_tight_fuzz.npz     VERDICT r4 item 1(ii): 220 problems, n <= 400, whose curvature bound is TIGHT -- kappa_bound drawn between 0.6 x and
                           1.0 x the maximum curvature of the box-only optimum (the regime of plateaus of adjacent active curvature rows, on
                           which round 4's block-pivoting phase ran on rounding noise).  Families: stadiums (two straights, two arcs: the
                           plateau case), random star-shaped rings, and the reference's own tracks (Berlin at the N = 333 of BASELINE config 2,
                           handling_track, rounded_rectangle) with the vehicle width varied.  Per problem: inputs, the oracle's alpha, its
                           curvature error, the number of active curvature / box rows, status_ref = 0 or 5 (the dense Goldfarb-Idnani reports
                           "constraints are inconsistent, no solution").
  oval_n2100.npz           VERDICT r4 item 2: the dense oracle ABOVE 2048 waypoints (the engine's long-ring route: tridiagonal sweeps on workspace
  oval_n2600.npz           vectors, the general interior point), config-5 generator; oval_n2600_kappa: curvature bound active at the optimum;
  oval_n2600_kappa.npz     shortest_path_n2100: tph.opt_shortest_path's QP at n = 2100 through the dense Goldfarb-Idnani
  shortest_path_n2100.npz

PARITY UNPINNED by the reference (tph / quadprog are not in /root/reference and not installable here): OUR oracle's outputs.
Run in the build container: `python scripts/make_golden_r5.py [fuzz|n2100|n2600|n2600k|sp2100 ...]`.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from global_racetrajectory_optimization_amd import synthetic  # noqa: E402
from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs  # noqa: E402
from oracle import qp_ref, tph_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def stadium(n, ls, r):
    """Two straights and two semicircles, n points equidistant in arclength (counter-clockwise)."""
    per = 2 * ls + 2 * np.pi * r
    xy = np.zeros((n, 2))
    for k, sk in enumerate(np.linspace(0.0, per, n, endpoint=False)):
        if sk < ls:
            xy[k] = (sk - ls / 2, -r)
        elif sk < ls + np.pi * r:
            th = (sk - ls) / r - np.pi / 2
            xy[k] = (ls / 2 + r * np.cos(th), r * np.sin(th))
        elif sk < 2 * ls + np.pi * r:
            xy[k] = (ls / 2 - (sk - ls - np.pi * r), r)
        else:
            th = (sk - 2 * ls - np.pi * r) / r + np.pi / 2
            xy[k] = (-ls / 2 + r * np.cos(th), r * np.sin(th))
    return xy


def star(n, rng):
    th = np.linspace(0.0, 2 * np.pi, n, endpoint=False)
    r = 40.0 + 6.0 * np.sin(3 * th + rng.uniform(0, 6)) + 3.0 * np.cos(5 * th + rng.uniform(0, 6))
    return np.column_stack((r * np.cos(th), r * np.sin(th)))


def prepared(xy):
    _, _, A, nv = tph_ref.calc_splines(np.vstack((xy, xy[0])))
    n = xy.shape[0]
    idx = np.arange(n - 1)
    sc = np.empty(n)
    sc[:-1] = -A[4 * idx + 2, 4 * idx + 5]
    sc[-1] = A[4 * n - 2, 1]
    return nv, A, sc


def fuzz():
    rng = np.random.default_rng(20260926)
    golden = {t: np.load(os.path.join(OUT, t + ".npz")) for t in ("berlin_2018_n333", "handling_track", "rounded_rectangle")}
    probs = []
    for k in range(220):
        fam = "stadium" if k < 90 else ("star" if k < 180 else "ref")
        if fam == "stadium":
            n = int(rng.integers(100, 401))
            xy = stadium(n, rng.uniform(60.0, 160.0), rng.uniform(25.0, 50.0))
            nv, A, sc = prepared(xy)
            w = np.full((n, 2), rng.uniform(3.0, 5.0)) if k % 3 else 3.0 + rng.uniform(0.0, 2.0, size=(n, 2))
            ref, w_veh = np.column_stack((xy, w)), 2.0
        elif fam == "star":
            n = int(rng.integers(60, 401))
            xy = star(n, rng)
            nv, A, sc = prepared(xy)
            ref, w_veh = np.column_stack((xy, 3.0 + rng.uniform(0.0, 1.5, size=(n, 2)))), 2.0
        else:
            g = golden[("berlin_2018_n333", "handling_track", "rounded_rectangle")[k % 3]]
            ref, nv, sc = g["reftrack"].copy(), g["normvec"], g["scaling"]
            A = cs.build_les_matrix(ref.shape[0], sc)
            w_veh = float(rng.uniform(1.6, 3.4))
        n = ref.shape[0]
        H, f, E, k_ref, aux = tph_ref.assemble_dense(ref, nv, A)
        G, h = tph_ref.constraints_dense(ref, E, k_ref, 1e9, w_veh)
        a_box = qp_ref.solve_qp_gi(H, f, G, h)
        kmax = float(np.max(np.abs(k_ref + E @ a_box)))
        kb = float(rng.uniform(0.6, 1.0)) * kmax
        info = {}
        G, h = tph_ref.constraints_dense(ref, E, k_ref, kb, w_veh)
        try:
            alpha = qp_ref.solve_qp_gi(H, f, G, h, info)
            status = 0
            curv_err = float(tph_ref.curv_error(alpha, aux))
            nk = int(np.sum(info["lagr"][2 * n:] > 0))
            nb = int(np.sum(info["lagr"][:2 * n] > 0))
        except ValueError as e:
            assert "inconsistent" in str(e)
            alpha, status, curv_err, nk, nb = np.zeros(n), 5, 0.0, 0, 0
        probs.append(dict(fam=fam, ref=ref, nv=nv, sc=sc, kb=kb, w_veh=w_veh, alpha=alpha, status=status, curv_err=curv_err, nk=nk, nb=nb, kmax=kmax))
        print("%3d %-8s n=%3d kb=%.4f (%.2f of the box optimum's %.4f) status %d, active curvature rows %d, box rows %d" % (
            k, fam, n, kb, kb / kmax, kmax, status, nk, nb), flush=True)
    off = np.concatenate(([0], np.cumsum([p["ref"].shape[0] for p in probs]))).astype(np.int64)
    np.savez_compressed(os.path.join(OUT, "kappa_tight_fuzz.npz"), offsets=off,
                        reftrack=np.concatenate([p["ref"] for p in probs]), normvec=np.concatenate([p["nv"] for p in probs]),
                        scaling=np.concatenate([p["sc"] for p in probs]), alpha=np.concatenate([p["alpha"] for p in probs]),
                        kappa_bound=np.array([p["kb"] for p in probs]), w_veh=np.array([p["w_veh"] for p in probs]),
                        status_ref=np.array([p["status"] for p in probs], dtype=np.int32), curv_error_max=np.array([p["curv_err"] for p in probs]),
                        n_active_kappa=np.array([p["nk"] for p in probs], dtype=np.int32), n_active_box=np.array([p["nb"] for p in probs], dtype=np.int32),
                        kappa_max_box_optimum=np.array([p["kmax"] for p in probs]))
    return dict(problems=len(probs), infeasible=int(sum(p["status"] == 5 for p in probs)),
                max_active_kappa=int(max(p["nk"] for p in probs)), mean_active_kappa=float(np.mean([p["nk"] for p in probs])))


def dense_pass(ref, nv, A, kappa_bound, w_veh):
    info = {}
    t0 = time.perf_counter()
    alpha, curv_err, I = tph_ref.opt_min_curv(ref, nv, A, kappa_bound, w_veh, return_internals=True,
                                              solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
    kkt = qp_ref.kkt_residuals(I["H"], I["f"], I["G"], I["h"], alpha)
    kap = I["k_ref"] + I["E"] @ alpha
    rec = dict(n=int(ref.shape[0]), gi_iters=[int(v) for v in info["iters"]], n_active=kkt["n_active"],
               kkt_stationarity=kkt["stationarity"], curv_error_max=float(curv_err), kappa_max=float(np.max(np.abs(kap))),
               n_active_kappa=int(np.count_nonzero(np.abs(np.abs(kap) - kappa_bound) < 1e-9)), seconds=time.perf_counter() - t0)
    return alpha, curv_err, rec


def long_oval(n, index, name, kappa_frac=None, w_veh=3.4):
    ref, nv, sc = synthetic.oval_batch(1, n=n, first=index, perturb_centreline=True)
    ref, nv, sc = ref[0], nv[0], sc[0]
    A = cs.build_les_matrix(n, sc)
    kb = 0.12
    if kappa_frac is not None:
        a0, _, r0 = dense_pass(ref, nv, A, 1e9, w_veh)
        kb = kappa_frac * r0["kappa_max"]
    alpha, curv_err, rec = dense_pass(ref, nv, A, kb, w_veh)
    rec.update(generator_index=index, kappa_bound=kb)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), input_sha256=np.array(sha(ref, nv, sc)), reftrack=ref, normvec=nv, scaling=sc,
                        alpha=alpha, curv_error_max=curv_err, kappa_bound=kb, w_veh=w_veh, generator_index=index, perturb_centreline=True)
    return rec


def shortest_path(n, index):
    ref, nv, sc = synthetic.oval_batch(1, n=n, first=index, perturb_centreline=True)
    ref, nv = ref[0], nv[0]
    t0 = time.perf_counter()
    alpha = tph_ref.opt_shortest_path(ref, nv, 3.4, solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h))
    np.savez_compressed(os.path.join(OUT, "shortest_path_n%d.npz" % n), input_sha256=np.array(sha(ref, nv)), reftrack=ref, normvec=nv,
                        alpha=alpha, w_veh=3.4, generator_index=index)
    return dict(n=n, seconds=time.perf_counter() - t0, path_length_sq=tph_ref.path_length_sq(ref, nv, alpha))


if __name__ == "__main__":
    what = sys.argv[1:] or ["fuzz", "n2100", "n2600", "n2600k", "sp2100"]
    summ_path = os.path.join(OUT, "SUMMARY_r5.json")
    summ = json.load(open(summ_path)) if os.path.exists(summ_path) else {}
    for w in what:
        t0 = time.time()
        if w == "fuzz":
            summ["kappa_tight_fuzz"] = fuzz()
        elif w == "n2100":
            summ["oval_n2100"] = long_oval(2100, 31, "oval_n2100")
        elif w == "n2600":
            summ["oval_n2600"] = long_oval(2600, 32, "oval_n2600")
        elif w == "n2600k":
            summ["oval_n2600_kappa"] = long_oval(2600, 33, "oval_n2600_kappa", kappa_frac=0.93)
        elif w == "sp2100":
            summ["shortest_path_n2100"] = shortest_path(2100, 34)
        print(w, "done in %.0f s" % (time.time() - t0), flush=True)
        json.dump(summ, open(summ_path, "w"), indent=1, sort_keys=True)
