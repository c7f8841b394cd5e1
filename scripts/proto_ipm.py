#!/usr/bin/env python
"""
Numpy model of the kernel's Mehrotra interior point + active-set identification on the box QP (dense H from the oracle's
assembly): used to try step rules / starting points / stopping rules before touching the HIP kernel.

  python scripts/proto_ipm.py [--n 2000] [--count 4]
"""
import argparse
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from global_racetrajectory_optimization_amd import synthetic                      # noqa: E402
from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs   # noqa: E402
from oracle import tph_ref                                                          # noqa: E402


def problem(ref, nv, sc, w_veh=3.4):
    n = ref.shape[0]
    A = cs.build_les_matrix(n, sc)
    H, f, E, kref, aux = tph_ref.assemble_dense(ref, nv, A)
    H = 0.5 * (H + H.T)
    lo = -(ref[:, 3] - 0.5 * w_veh)
    hi = ref[:, 2] - 0.5 * w_veh
    return H, f, lo, hi


def exact_active_set(H, f, lo, hi, st0, max_iter=100):
    """Block principal pivoting from a guess; returns (x, state, iterations)."""
    n = len(f)
    st = st0.copy()
    for it in range(1, max_iter + 1):
        F = st == 0
        x = np.where(st < 0, lo, hi).astype(float)
        x[F] = 0.0
        rhs = -(f[F] + H[np.ix_(F, ~F)] @ x[~F])
        x[F] = sla.cho_solve(sla.cho_factor(H[np.ix_(F, F)]), rhs)
        g = H @ x + f
        bad_p = F & ((x < lo - 1e-12) | (x > hi + 1e-12))
        bad_d = ((st < 0) & (g < -1e-10 * np.abs(f).max())) | ((st > 0) & (g > 1e-10 * np.abs(f).max()))
        if not bad_p.any() and not bad_d.any():
            return x, st, it
        st = st.copy()
        st[bad_p & (x < lo)] = -1
        st[bad_p & (x > hi)] = 1
        st[bad_d] = 0
    return x, st, max_iter


def ipm(H, f, lo, hi, tol=1e-10, gamma=0.995, adaptive=False, start="centre", stop_rule=None, max_iter=60, log=False):
    n = len(f)
    x = 0.5 * (lo + hi)
    g = H @ x + f
    zscale = np.abs(g).max()
    wmean = (hi - lo).mean()
    zl = np.full(n, zscale)
    zu = np.full(n, zscale)
    if start == "mehrotra":
        # complementarity-balanced start: z_i = mu0 / s_i with mu0 from the gradient scale
        sl, su = x - lo, hi - x
        mu0 = zscale * wmean * 0.5 * 0.1
        zl = mu0 / sl
        zu = mu0 / su
    hist = []
    for it in range(1, max_iter + 1):
        g = H @ x + f
        sl, su = x - lo, hi - x
        mu = (sl @ zl + su @ zu) / (2 * n)
        rd = np.abs(g - zl + zu).max()
        hist.append((mu / (zscale * wmean), rd / zscale))
        if stop_rule is not None and stop_rule(it, mu / (zscale * wmean), rd / zscale, x, zl, zu, lo, hi, zscale):
            return x, zl, zu, it - 1, hist
        if mu < tol * zscale * wmean and rd < tol * zscale:
            return x, zl, zu, it - 1, hist
        sig = zl / sl + zu / su
        cf = sla.cho_factor(H + np.diag(sig))
        dxa = sla.cho_solve(cf, -g)
        dzla = -zl - zl * dxa / sl
        dzua = -zu + zu * dxa / su

        def maxstep(dx, dzl, dzu):
            a = np.inf
            m = dx < 0
            if m.any(): a = min(a, (-sl[m] / dx[m]).min())
            m = dx > 0
            if m.any(): a = min(a, (su[m] / dx[m]).min())
            m = dzl < 0
            if m.any(): a = min(a, (-zl[m] / dzl[m]).min())
            m = dzu < 0
            if m.any(): a = min(a, (-zu[m] / dzu[m]).min())
            return a
        # predictor uses separate primal / dual lengths like the kernel
        ap = 1.0
        m = dxa < 0
        if m.any(): ap = min(ap, (-sl[m] / dxa[m]).min())
        m = dxa > 0
        if m.any(): ap = min(ap, (su[m] / dxa[m]).min())
        ad = 1.0
        m = dzla < 0
        if m.any(): ad = min(ad, (-zl[m] / dzla[m]).min())
        m = dzua < 0
        if m.any(): ad = min(ad, (-zu[m] / dzua[m]).min())
        mua = ((sl + ap * dxa) @ (zl + ad * dzla) + (su - ap * dxa) @ (zu + ad * dzua)) / (2 * n)
        smu = (mua / mu) ** 3 * mu
        rhs = -g + (smu - dxa * dzla) / sl - (smu + dxa * dzua) / su
        dx = sla.cho_solve(cf, rhs)
        dzl = (-sl * zl + smu - dxa * dzla - zl * dx) / sl
        dzu = (-su * zu + smu + dxa * dzua + zu * dx) / su
        amax = maxstep(dx, dzl, dzu)
        gm = gamma
        if adaptive:
            gm = max(gamma, 1.0 - 10.0 * mu / (zscale * wmean))
            gm = min(gm, 1.0 - 1e-9)
        a = min(1.0, gm * amax)
        x = x + a * dx
        zl = zl + a * dzl
        zu = zu + a * dzu
        if log:
            print("   it %2d mu %.2e rd %.2e a %.4f sigma %.2e" % (it, hist[-1][0], hist[-1][1], a, (mua / mu) ** 3))
    return x, zl, zu, max_iter, hist


def identify(x, zl, zu, lo, hi, zscale):
    wdt = hi - lo
    sl, su = x - lo, hi - x
    st = np.zeros(len(x), dtype=np.int8)
    st[sl * zscale < zl * wdt] = -1
    st[(su * zscale < zu * wdt) & (st == 0)] = 1
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--count", type=int, default=3)
    ap.add_argument("--first", type=int, default=0)
    args = ap.parse_args()
    ref_b, nv_b, sc_b = synthetic.oval_batch(args.count, n=args.n, first=args.first)
    variants = {
        "kernel (tol 1e-10, 0.995)": dict(),
        "adaptive step": dict(adaptive=True),
        "tol 1e-8": dict(tol=1e-8),
        "tol 1e-8 adaptive": dict(tol=1e-8, adaptive=True),
        "tol 1e-6 adaptive": dict(tol=1e-6, adaptive=True),
        "mehrotra start": dict(start="mehrotra"),
        "mehrotra start adaptive": dict(start="mehrotra", adaptive=True),
    }
    for k in range(args.count):
        H, f, lo, hi = problem(ref_b[k], nv_b[k], sc_b[k])
        print("problem %d" % k)
        for name, kw in variants.items():
            x, zl, zu, it, hist = ipm(H, f, lo, hi, **kw)
            zscale = np.abs(H @ (0.5 * (lo + hi)) + f).max()
            st = identify(x, zl, zu, lo, hi, zscale)
            xs, st2, asit = exact_active_set(H, f, lo, hi, st)
            print("  %-28s ipm %2d  as %2d  guess wrong %3d  active %3d" % (name, it, asit, int((st != st2).sum()), int((st2 != 0).sum())))


if __name__ == "__main__":
    main()
