"""
Round-3 additions to tests/golden/ (the existing fixtures are NOT regenerated, they stay bitwise as committed):

  berlin_2018_iqp.npz, modena_2019_iqp.npz
      the reference's DEFAULT flow on its shipped tracks [REF main_globaltraj.py:273-284, params/racecar.ini:72-74]:
      tph_ref.iqp_handler (dense 4N x 4N re-linearisation + dense Goldfarb-Idnani with all 4N rows every pass) from the
      reftrack / normals / scalings stored in berlin_2018.npz / modena_2019.npz: end state + per-pass trace.
  berlin_2018_n333.npz
      BASELINE config 2's second size: spline_approximation(stepsize_reg = 7.0) on inputs/tracks/berlin_2018.csv (not a
      subsampling of the N = 776 ring), first pass through the dense oracle, second route (BVLS n/a at this size: TRF), KKT.
  oval_n2000_w1.npz, oval_n2000_w2.npz
      two further width seeds (generator index 1, 2) of the bench workload, N = 2000, first pass through the dense oracle.
  oval_n2000_c5.npz, oval_n2000_c9.npz
      two tracks of BASELINE config 5's generator (perturb_centreline = True, generator index 5 and 9), N = 2000.

PARITY UNPINNED by the reference (it ships no vectors; tph / quadprog not installable): OUR oracle's outputs, each with a
KKT certificate and, for the first passes, an independent second solution route (lsq_linear(trf) on the dense E).

Run in the BUILD container (the Berlin N = 333 fixture reads /root/reference/inputs/tracks).  `python scripts/make_golden_r3.py
[names...]`; about 25 minutes of 8 cores for everything.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from global_racetrajectory_optimization_amd import synthetic  # noqa: E402
from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs  # noqa: E402
from global_racetrajectory_optimization_amd.trajectory_planning_helpers import spline_approximation as sa  # noqa: E402
from oracle import qp_ref, tph_ref  # noqa: E402

REF_TRACKS = "/root/reference/inputs/tracks"
OUT = os.path.join(ROOT, "tests", "golden")
KAPPA_BOUND, W_VEH = 0.12, 3.4
STEPSIZE, ITERS_MIN, CURV_ERR_ALLOWED = 3.0, 3, 0.01


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return h.hexdigest()


def first_pass(ref, nv, A, second_route=True):
    info = {}
    t0 = time.perf_counter()
    alpha, curv_err, I = tph_ref.opt_min_curv(ref, nv, A, KAPPA_BOUND, W_VEH, return_internals=True,
                                              solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
    t_first = time.perf_counter() - t0
    kkt = qp_ref.kkt_residuals(I["H"], I["f"], I["G"], I["h"], alpha)
    rec = dict(n=int(ref.shape[0]), gi_iters=[int(v) for v in info["iters"]], n_active=kkt["n_active"],
               kkt_stationarity=kkt["stationarity"], curv_error_max=curv_err,
               kappa_max=float(np.max(np.abs(I["k_ref"] + I["E"] @ alpha))), seconds_first_pass=t_first)
    if second_route:
        lo, hi = -(ref[:, 3] - W_VEH / 2), ref[:, 2] - W_VEH / 2
        t0 = time.perf_counter()
        a2 = qp_ref.solve_box_second_route(I["E"], I["k_ref"], lo, hi)
        rec["second_route"] = "lsq_linear(trf) on dense E"
        rec["second_route_max_diff"] = float(np.max(np.abs(a2 - alpha)))
        rec["seconds_second_route"] = time.perf_counter() - t0
    return alpha, curv_err, rec


def iqp_fixture(name):
    g = np.load(os.path.join(OUT, name + ".npz"))
    ref, nv, sc = g["reftrack"], g["normvec"], g["scaling"]
    A = cs.build_les_matrix(ref.shape[0], sc)
    trace = []
    t0 = time.perf_counter()
    a_iqp, ref_iqp, nv_iqp = tph_ref.iqp_handler(ref, nv, A, KAPPA_BOUND, W_VEH, STEPSIZE, ITERS_MIN, CURV_ERR_ALLOWED, trace=trace)
    t_iqp = time.perf_counter() - t0
    out = dict(input_sha256=np.array(sha(ref, nv, sc)), iqp_alpha=a_iqp, iqp_reftrack=ref_iqp, iqp_normvec=nv_iqp,
               iqp_n=np.array([t["n"] for t in trace]), iqp_curv_err=np.array([t["curv_error_max"] for t in trace]),
               kappa_bound=KAPPA_BOUND, w_veh=W_VEH, stepsize_interp=STEPSIZE, iters_min=ITERS_MIN,
               curv_error_allowed=CURV_ERR_ALLOWED)
    for k, t in enumerate(trace):
        out["iqp_pass%d_alpha" % (k + 1)] = t["alpha"]
    np.savez_compressed(os.path.join(OUT, name + "_iqp.npz"), **out)
    return dict(n=int(ref.shape[0]), iqp_n=[int(t["n"]) for t in trace], iqp_curv_err=[float(t["curv_error_max"]) for t in trace],
                seconds_iqp=t_iqp)


def berlin_n333():
    trk = np.loadtxt(os.path.join(REF_TRACKS, "berlin_2018.csv"), comments="#", delimiter=",")
    ref = sa.spline_approximation(trk, k_reg=3, s_reg=10, stepsize_prep=1.0, stepsize_reg=7.0)
    n = ref.shape[0]
    path_cl = np.vstack((ref[:, :2], ref[0, :2]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    idx = np.arange(n - 1)
    sc = np.empty(n)
    sc[:-1] = -A[4 * idx + 2, 4 * idx + 5]
    sc[-1] = A[4 * n - 2, 1]
    alpha, curv_err, rec = first_pass(ref, nv, A)
    np.savez_compressed(os.path.join(OUT, "berlin_2018_n333.npz"), input_sha256=np.array(sha(ref, nv, sc)), reftrack=ref, normvec=nv,
                        scaling=sc, alpha=alpha, curv_error_max=curv_err, kappa_bound=KAPPA_BOUND, w_veh=W_VEH, stepsize_reg=7.0)
    return rec


def oval(index, perturb):
    ref, nv, sc = synthetic.oval_batch(1, n=2000, first=index, perturb_centreline=perturb)
    ref, nv, sc = ref[0], nv[0], sc[0]
    A = cs.build_les_matrix(ref.shape[0], sc)
    alpha, curv_err, rec = first_pass(ref, nv, A)
    rec.update(generator_index=index, perturb_centreline=perturb)
    name = "oval_n2000_%s%d" % ("c" if perturb else "w", index)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), input_sha256=np.array(sha(ref, nv, sc)), reftrack=ref, normvec=nv, scaling=sc,
                        alpha=alpha, curv_error_max=curv_err, kappa_bound=KAPPA_BOUND, w_veh=W_VEH, generator_index=index,
                        perturb_centreline=perturb)
    return rec


JOBS = {
    "berlin_2018_n333": berlin_n333,
    "modena_2019_iqp": lambda: iqp_fixture("modena_2019"),
    "berlin_2018_iqp": lambda: iqp_fixture("berlin_2018"),
    "oval_n2000_w1": lambda: oval(1, False),
    "oval_n2000_w2": lambda: oval(2, False),
    "oval_n2000_c5": lambda: oval(5, True),
    "oval_n2000_c9": lambda: oval(9, True),
}


def main():
    qp_ref.build()
    names = sys.argv[1:] or list(JOBS)
    path = os.path.join(OUT, "SUMMARY_r3.json")
    summary = json.load(open(path)) if os.path.exists(path) else {}
    for name in names:
        rec = JOBS[name]()
        summary[name] = rec
        print(name, rec, flush=True)
        with open(path, "w") as fh:
            json.dump(summary, fh, indent=1)


if __name__ == "__main__":
    main()
