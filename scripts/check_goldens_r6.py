"""
Round 6: every fixture of tests/golden/ RE-SOLVED by the oracle after oracle/gi_dense.c was rewritten to qpgen2's rule set (quadprog's own
Goldfarb-Idnani: no exclusion list, unconditional adds, `inconsistent` iff t1 = inf and z = 0, vsmall) -- the committed goldens are NOT
regenerated; this script reports how far the new oracle's answers are from them and writes tests/golden/CHECK_r6.json.

What it reports per fixture:
  d_alpha        max |alpha(new oracle) - alpha(committed)|  (metres)
  same_active    the new oracle's final active set (quadprog's iact) is the set of rows the committed alpha sits on (1e-7 m / 1e-9 1/m)
  iters          quadprog's `iters` pair: main iterations (= full steps + 1), drops
  d_vertex_new / d_vertex_old   (box-only fixtures) distance of both alphas to the VERTEX of that active set computed independently of any
                 QP arithmetic: the equality-constrained least-squares problem on the free columns of the dense E, solved by QR with two
                 rounds of refinement whose residuals are accumulated in extended precision (np.longdouble).  Two solves of a QP whose
                 reduced Hessian has condition 1e9 .. 1e12 can differ by eps x cond and both be "right"; the vertex says by how much each is.
  kappa_tight_fuzz  the 220 verdicts (0 / "constraints are inconsistent") and alphas.

PARITY UNPINNED by the reference (tph / quadprog are third-party, absent from /root/reference, not installable here): this is OUR oracle
checked against OUR committed outputs.  Run in the build container: `python scripts/check_goldens_r6.py [names ...]` (about 25 min for all).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import qp_ref, tph_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

FIRST_PASS = ["rounded_rectangle", "handling_track", "modena_2019", "berlin_2018", "berlin_2018_n333",
              "oval_n2000", "oval_n2000_w1", "oval_n2000_w2", "oval_n2000_w3", "oval_n2000_w7", "oval_n2000_w11",
              "oval_n2000_c5", "oval_n2000_c9", "oval_n2000_c13", "oval_n2000_c21", "oval_n2000_kappa",
              "oval_n2100", "oval_n2600", "oval_n2600_kappa"]
UNIT_SCALING = ["iqp_pass2_oval5", "iqp_pass3_oval3", "iqp_pass3_oval629", "iqp_pass3_oval9"]
IQP_CHAINS = [("rounded_rectangle", "rounded_rectangle"), ("handling_track", "handling_track"), ("modena_2019_iqp", "modena_2019"),
              ("berlin_2018_iqp", "berlin_2018"), ("oval_n2000", "oval_n2000")]


def vertex_by_refined_least_squares(E, k_ref, alpha, lo, hi, tol=1e-7):
    """The minimiser of 1/2 |E a + 2 k_ref|^2 with the rows `alpha` sits on held at their bounds: QR on the free columns + refinement with
    extended-precision residuals.  Returns (vertex, n_active)."""
    at_hi = np.abs(alpha - hi) <= tol
    at_lo = np.abs(alpha - lo) <= tol
    x = np.where(at_hi, hi, np.where(at_lo, lo, 0.0))
    free = ~(at_hi | at_lo)
    Ef = E[:, free]
    Q, R = np.linalg.qr(Ef)
    El = E.astype(np.longdouble)
    b = (-tph_ref.F_SCALE * k_ref).astype(np.longdouble) - El[:, ~free] @ x[~free].astype(np.longdouble)
    xf = np.zeros(int(free.sum()), dtype=np.longdouble)
    for _ in range(4):
        res = (b - El[:, free] @ xf).astype(np.float64)
        import scipy.linalg
        dx = scipy.linalg.solve_triangular(R, Q.T @ res)
        xf = xf + dx
    x[free] = xf.astype(np.float64)
    return x, int((~free).sum())


def one_qp(name, ref, nv, A, kappa_bound, w_veh, alpha_gold, curv_gold):
    t0 = time.time()
    H, f, E, k_ref, aux = tph_ref.assemble_dense(ref, nv, A)
    G, h = tph_ref.constraints_dense(ref, E, k_ref, kappa_bound, w_veh)
    info = {}
    alpha = qp_ref.solve_qp_gi(H, f, G, h, info)
    err = tph_ref.curv_error(alpha, aux)
    n = ref.shape[0]
    iact = set(int(i) for i in info["iact"])
    s_gold = h - G @ alpha_gold
    on_gold = set(int(i) for i in np.where(s_gold <= np.where(np.arange(4 * n) < 2 * n, 1e-7, 1e-9))[0])
    # a row with a zero multiplier may sit on its bound without being in quadprog's working set: compare on the rows that carry force
    strong = set(int(i) for i in iact if info["lagr"][i] > 1e-12 * np.max(np.abs(f)))
    rec = dict(n=int(n), d_alpha=float(np.max(np.abs(alpha - alpha_gold))), d_curv_error=float(abs(err - curv_gold)),
               iters=[int(info["iters"][0]), int(info["iters"][1])], n_active=len(iact), n_active_kappa=sum(1 for i in iact if i >= 2 * n),
               same_active=bool(strong <= on_gold and on_gold <= iact | set(np.where(h - G @ alpha <= 1e-7)[0].tolist())),
               kkt_stationarity=qp_ref.kkt_residuals(H, f, G, h, alpha)["stationarity"])
    if rec["n_active_kappa"] == 0:
        lo, hi = -(ref[:, 3] - w_veh / 2), ref[:, 2] - w_veh / 2
        v, na = vertex_by_refined_least_squares(E, k_ref, alpha, lo, hi)
        rec.update(d_vertex_new=float(np.max(np.abs(alpha - v))), d_vertex_old=float(np.max(np.abs(alpha_gold - v))), vertex_rows=na)
    rec["seconds"] = time.time() - t0
    print(name, json.dumps(rec), flush=True)
    return rec


def scaling_matrix(ref, unit):
    path_cl = np.vstack((ref[:, :2], ref[0, :2]))
    _, _, A, nv = tph_ref.calc_splines(path_cl, use_dist_scaling=not unit)
    return A, nv


def main(argv):
    want = set(argv)
    out_path = os.path.join(GOLD, "CHECK_r6.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}

    def sel(key):
        return not want or key in want

    for name in FIRST_PASS + UNIT_SCALING:
        if not sel(name):
            continue
        g = np.load(os.path.join(GOLD, name + ".npz"))
        A, nv = scaling_matrix(g["reftrack"], name in UNIT_SCALING)
        assert np.max(np.abs(nv - g["normvec"])) < 1e-9, name
        out[name] = one_qp(name, g["reftrack"], g["normvec"], A, float(g["kappa_bound"]), float(g["w_veh"]), g["alpha"], float(g["curv_error_max"]))
        json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)

    for fix, base in IQP_CHAINS:
        key = "iqp_chain_" + base
        if not sel(key):
            continue
        t0 = time.time()
        g = np.load(os.path.join(GOLD, fix + ".npz"))
        b = np.load(os.path.join(GOLD, base + ".npz"))
        A, nv = scaling_matrix(b["reftrack"], False)
        trace = []
        al, rt, nvn = tph_ref.iqp_handler(b["reftrack"], b["normvec"], A, 0.12, 3.4, 3.0, iters_min=3, curv_error_allowed=0.01, trace=trace)
        rec = dict(passes=len(trace), n=[int(t["n"]) for t in trace], same_shape=bool(al.shape == g["iqp_alpha"].shape))
        if rec["same_shape"]:
            rec.update(d_alpha=float(np.max(np.abs(al - g["iqp_alpha"]))), d_reftrack=float(np.max(np.abs(rt - g["iqp_reftrack"]))),
                       d_curv_err=float(np.max(np.abs(np.array([t["curv_error_max"] for t in trace]) - g["iqp_curv_err"]))))
        rec["seconds"] = time.time() - t0
        print(key, json.dumps(rec), flush=True)
        out[key] = rec
        json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)

    if sel("shortest_path"):
        z = np.load(os.path.join(GOLD, "shortest_path.npz"))
        rec = {}
        for base in ("rounded_rectangle", "handling_track", "modena_2019", "berlin_2018"):
            b = np.load(os.path.join(GOLD, base + ".npz"))
            rec[base] = float(np.max(np.abs(tph_ref.opt_shortest_path(b["reftrack"], b["normvec"], float(z["w_veh"])) - z[base + "_alpha"])))
        g = np.load(os.path.join(GOLD, "shortest_path_n2100.npz"))
        rec["n2100"] = float(np.max(np.abs(tph_ref.opt_shortest_path(g["reftrack"], g["normvec"], float(g["w_veh"])) - g["alpha"])))
        print("shortest_path", json.dumps(rec), flush=True)
        out["shortest_path"] = rec
        json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)

    if sel("kappa_tight_fuzz"):
        t0 = time.time()
        z = np.load(os.path.join(GOLD, "kappa_tight_fuzz.npz"))
        off = z["offsets"]
        changed, worst, worst_k, n_inc = [], 0.0, -1, 0
        for k in range(len(off) - 1):
            ref, nv = z["reftrack"][off[k]:off[k + 1]], z["normvec"][off[k]:off[k + 1]]
            A, _ = scaling_matrix(ref, False)
            try:
                al, _ = tph_ref.opt_min_curv(ref, nv, A, float(z["kappa_bound"][k]), float(z["w_veh"][k]))
                st = 0
            except ValueError as e:
                assert "inconsistent" in str(e)
                st, al = 5, None
                n_inc += 1
            if st != int(z["status_ref"][k]):
                changed.append(k)
            elif st == 0:
                d = float(np.max(np.abs(al - z["alpha"][off[k]:off[k + 1]])))
                if d > worst:
                    worst, worst_k = d, k
        rec = dict(problems=int(len(off) - 1), inconsistent=n_inc, verdicts_changed=changed, worst_d_alpha=worst, worst_problem=worst_k,
                   seconds=time.time() - t0)
        print("kappa_tight_fuzz", json.dumps(rec), flush=True)
        out["kappa_tight_fuzz"] = rec
        json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
