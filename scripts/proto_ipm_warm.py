#!/usr/bin/env python
"""
Numpy model for VERDICT r5 item 6: a warm-start FALLBACK cheaper than the cold path.  When the warm-started exchange of an IQP pass runs out of its
12 rounds, the kernel starts the interior point from the box centre (10-11 iterations).  Here: the interior point started from the exchange's last
iterate instead -- pushed inside its box by delta x width, multipliers from the gradient there with a floor mu0 / slack -- on the third-pass
regression instances (tests/golden/iqp_pass3_oval*.npz: dozens of barely active bounds), the exchange's iterate emulated by solving the
equality-constrained problem on the optimal working set with a handful of rows flipped.

  python scripts/proto_ipm_warm.py
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proto_ipm import exact_active_set, identify                        # noqa: E402
from oracle import tph_ref                                               # noqa: E402


def ipm_from(H, f, lo, hi, x, zl, zu, tol=1e-10, gamma=0.995, max_iter=60):
    n = len(f)
    xc = 0.5 * (lo + hi)
    zscale = np.abs(H @ xc + f).max()
    wmean = (hi - lo).mean()
    for it in range(1, max_iter + 1):
        g = H @ x + f
        sl, su = x - lo, hi - x
        mu = (sl @ zl + su @ zu) / (2 * n)
        rd = np.abs(g - zl + zu).max()
        if mu < tol * zscale * wmean and rd < tol * zscale:
            return x, zl, zu, it - 1
        sig = zl / sl + zu / su
        cf = sla.cho_factor(H + np.diag(sig))
        rp = -(g - zl + zu) - zl + zu          # = -g (the complementarity terms fold in)
        dxa = sla.cho_solve(cf, -g)
        dzla = -zl - zl * dxa / sl
        dzua = -zu + zu * dxa / su

        def lengths(dx, dzl, dzu):
            a = np.inf
            for (v, d) in ((sl, dx), (su, -dx), (zl, dzl), (zu, dzu)):
                m = d < 0
                if m.any():
                    a = min(a, (-v[m] / d[m]).min())
            return a
        aa = min(1.0, lengths(dxa, dzla, dzua))
        mua = ((sl + aa * dxa) @ (zl + aa * dzla) + (su - aa * dxa) @ (zu + aa * dzua)) / (2 * n)
        smu = (mua / mu) ** 3 * mu
        rhs = -g + (smu - dxa * dzla) / sl - (smu + dxa * dzua) / su
        dx = sla.cho_solve(cf, rhs)
        dzl = (-sl * zl + smu - dxa * dzla - zl * dx) / sl
        dzu = (-su * zu + smu + dxa * dzua + zu * dx) / su
        a = min(1.0, gamma * lengths(dx, dzl, dzu))
        x, zl, zu = x + a * dx, zl + a * dzl, zu + a * dzu
    return x, zl, zu, max_iter


def main():
    rng = np.random.default_rng(7)
    for name in ("iqp_pass3_oval3", "iqp_pass3_oval629", "iqp_pass3_oval9", "iqp_pass2_oval5"):
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        ref, nv = g["reftrack"], g["normvec"]
        n = ref.shape[0]
        A = tph_ref.calc_splines(np.vstack((ref[:, :2], ref[0, :2])), use_dist_scaling=False)[2]
        H, f, E, kref, _ = tph_ref.assemble_dense(ref, nv, A)
        H = 0.5 * (H + H.T)
        lo, hi = -(ref[:, 3] - 1.7), ref[:, 2] - 1.7
        xs = g["alpha"]
        st_opt = np.where(np.abs(xs - lo) < 1e-9, -1, np.where(np.abs(xs - hi) < 1e-9, 1, 0)).astype(np.int8)
        zscale = np.abs(H @ (0.5 * (lo + hi)) + f).max()
        wmean = (hi - lo).mean()
        n = len(f)
        # cold
        xc = 0.5 * (lo + hi)
        _, _, _, it_cold = ipm_from(H, f, lo, hi, xc, np.full(n, zscale), np.full(n, zscale))
        print("%s: n %d active %d; cold interior point %d iterations" % (name, n, int((st_opt != 0).sum()), it_cold), flush=True)
        for flips in (4, 12):
            st = st_opt.copy()
            act = np.where(st_opt != 0)[0]
            st[rng.choice(act, flips // 2, replace=False)] = 0                     # active rows released
            free = np.where(st_opt == 0)[0]
            nb = free[np.argsort(np.minimum(xs[free] - lo[free], hi[free] - xs[free]))[:flips]]      # rows closest to a bound pinned there
            for i in nb[:flips // 2]:
                st[i] = -1 if xs[i] - lo[i] < hi[i] - xs[i] else 1
            F = st == 0
            x = np.where(st < 0, lo, hi).astype(float)
            x[F] = 0.0
            x[F] = sla.cho_solve(sla.cho_factor(H[np.ix_(F, F)]), -(f[F] + H[np.ix_(F, ~F)] @ x[~F]))
            gx = H @ x + f
            for delta in (1e-2, 1e-3, 1e-4):
                for mu0 in (1e-3, 1e-5):
                    xw = np.minimum(np.maximum(x, lo + delta * (hi - lo)), hi - delta * (hi - lo))
                    gw = H @ xw + f
                    floor = mu0 * zscale * wmean
                    zl = np.maximum(gw, 0.0) + floor / (xw - lo)
                    zu = np.maximum(-gw, 0.0) + floor / (hi - xw)
                    _, _, _, it = ipm_from(H, f, lo, hi, xw, zl, zu)
                    print("   %2d rows flipped, iterate %.1e m from the optimum: warm start delta %.0e mu0 %.0e -> %2d iterations"
                          % (flips, np.max(np.abs(x - xs)), delta, mu0, it), flush=True)


if __name__ == "__main__":
    main()
