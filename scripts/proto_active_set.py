#!/usr/bin/env python
"""
Numpy model of the kernel's active-set phase (block principal pivoting with the Kim-Park / Judice-Pires backup) on dense H,
used to compare exchange rules before touching the HIP kernel.  Two rules for the rows that leave their box in a
full-exchange round:

  stretch        pin every such row (the first version of the kernel)
  neighbourhood  pin only the row that leaves the box furthest among the rows R either side (MCQ_AS_WINDOW in the kernel)

on the degenerate third-IQP-pass fixtures (tests/golden/iqp_pass3_oval*.npz) and on first-pass problems of the bench
workload, from the interior point's guess at several tolerances.  Prints interior-point iterations + pivoting rounds.

  python scripts/proto_active_set.py [--first-pass 3]        # a few minutes (dense Cholesky, N = 2000)
"""
import argparse
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import proto_ipm                                                                   # noqa: E402
from global_racetrajectory_optimization_amd import synthetic                      # noqa: E402
from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs   # noqa: E402
from oracle import tph_ref                                                          # noqa: E402


def active_set(H, f, lo, hi, st0, cap, window):
    """The kernel's loop; window = 0: 'stretch' rule, window = R > 0: 'neighbourhood' rule.  Returns (x, rounds, converged)."""
    n = len(f)
    st = st0.copy()
    fscale = np.abs(f).max()
    best, pcnt = 2 * n + 1, 3
    for it in range(1, cap + 1):
        F = st == 0
        x = np.where(st < 0, lo, hi).astype(float)
        x[F] = sla.cho_solve(sla.cho_factor(H[np.ix_(F, F)]), -(f[F] + H[np.ix_(F, ~F)] @ x[~F]))
        g = H @ x + f
        v = np.zeros(n, dtype=int)
        v[F & (x < lo - 1e-10)] = -1
        v[F & (x > hi + 1e-10)] = 1
        v[(st == -1) & (g < -1e-10 * fscale)] = 2
        v[(st == 1) & (g > 1e-10 * fscale)] = 2
        pv = np.where(np.abs(v) == 1, np.maximum(lo - x, x - hi), 0.0)
        nv = int((v != 0).sum())
        if nv == 0:
            return x, it, True
        if nv < best:
            best, pcnt, full = nv, 3, True
        elif pcnt > 0:
            pcnt, full = pcnt - 1, True
        else:
            full = False
        idx = np.nonzero(v)[0]
        if not full:
            idx = idx[-1:]
        for i in idx:
            if v[i] == 2:
                st[i] = 0
            elif not full or window == 0 or all(pv[(i - d) % n] < pv[i] and pv[(i + d) % n] <= pv[i] for d in range(1, window + 1)):
                st[i] = v[i]
    return x, cap, False


def study(name, H, f, lo, hi):
    cells = []
    for tol in (1e-10, 1e-8, 1e-6):
        x, zl, zu, it, _ = proto_ipm.ipm(H, f, lo, hi, tol=tol)
        zscale = np.abs(H @ (0.5 * (lo + hi)) + f).max()
        st = proto_ipm.identify(x, zl, zu, lo, hi, zscale)
        rounds = []
        for window in (0, 8):
            _, r, ok = active_set(H, f, lo, hi, st, 60, window)
            rounds.append("%2d%s" % (r, " " if ok else "!"))
        cells.append("tol %.0e: ipm %2d, rounds stretch %s neighbourhood %s" % (tol, it, rounds[0], rounds[1]))
    print("%-22s %s" % (name, " | ".join(cells)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first-pass", type=int, default=3)
    args = ap.parse_args()
    for name in ("iqp_pass3_oval629", "iqp_pass3_oval3"):
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        ref, nv = z["reftrack"], z["normvec"]
        n = ref.shape[0]
        H, f, _, _, _ = tph_ref.assemble_dense(ref, nv, cs.build_les_matrix(n, np.ones(n)))
        study(name, 0.5 * (H + H.T), f, -(ref[:, 3] - 1.7), ref[:, 2] - 1.7)
    ref_b, nv_b, sc_b = synthetic.oval_batch(args.first_pass, n=2000)
    for k in range(args.first_pass):
        study("first pass, oval %d" % k, *proto_ipm.problem(ref_b[k], nv_b[k], sc_b[k]))


if __name__ == "__main__":
    main()
