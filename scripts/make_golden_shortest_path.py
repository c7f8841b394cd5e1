"""
Generates tests/golden/shortest_path.npz: the shortest-path QP on the four golden tracks.

Inputs are the reftrack / normvec arrays already held by tests/golden/<track>.npz (made by scripts/make_golden.py from the
reference's track CSVs), the vehicle width is the reference's default for this mode [REF params/racecar.ini:65].
PARITY UNPINNED by the reference, as for the minimum-curvature vectors: the alphas
come from OUR restatement (oracle/tph_ref.opt_shortest_path + oracle/gi_dense.c) and are pinned by an independent route
(scipy BVLS on the Cholesky least-squares form, tracks with n <= 300) plus a KKT certificate, both recorded.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import qp_ref, tph_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
W_VEH = 3.4


def polygon_length(ref, nv, alpha):
    p = ref[:, :2] + alpha[:, None] * nv
    return float(np.sum(np.linalg.norm(np.roll(p, -1, axis=0) - p, axis=1)))


def main():
    qp_ref.build()
    out, summary = {}, {}
    for name in ("rounded_rectangle", "handling_track", "modena_2019", "berlin_2018"):
        z = np.load(os.path.join(OUT, name + ".npz"))
        ref, nv = z["reftrack"], z["normvec"]
        n = ref.shape[0]
        alpha, I = tph_ref.opt_shortest_path(ref, nv, W_VEH, return_internals=True)
        kkt = qp_ref.kkt_residuals(I["H"], I["f"], I["G"], I["h"], alpha)
        rec = dict(n=n, n_active=kkt["n_active"], kkt_stationarity=kkt["stationarity"],
                   length_ref=polygon_length(ref, nv, np.zeros(n)), length_opt=polygon_length(ref, nv, alpha))
        if n <= 300:
            from scipy.optimize import lsq_linear
            R = np.linalg.cholesky(I["H"]).T                              # H = R'R
            b = -np.linalg.solve(R.T, I["f"])                             # 1/2 |R a - b|^2 = 1/2 a'Ha + f'a + const
            res = lsq_linear(R, b, bounds=(-I["h"][n:], I["h"][:n]), method="bvls", tol=1e-15, max_iter=50 * n)
            rec["bvls_max_diff"] = float(np.max(np.abs(res.x - alpha)))
        out[name + "_alpha"] = alpha
        summary[name] = rec
        print(name, rec)
    out["w_veh"] = W_VEH
    np.savez_compressed(os.path.join(OUT, "shortest_path.npz"), **out)
    with open(os.path.join(OUT, "SUMMARY_shortest_path.json"), "w") as fh:
        json.dump(summary, fh, indent=1)


if __name__ == "__main__":
    main()
