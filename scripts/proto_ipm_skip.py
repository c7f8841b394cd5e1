#!/usr/bin/env python
"""
Numpy model for VERDICT r5 item 4(b): ADAPTIVE corrector skipping in the Mehrotra interior point of the box QP.  Where the affine (predictor)
step is nearly full (primal and dual step lengths >= thr), sigma = (mu_aff / mu)^3 is tiny and the corrector mostly re-solves for the same
direction: take the affine step itself and skip the corrector's solve (in the kernel: a solve with its own chains, 496 B per waypoint, plus the
second step-length pass).  Reports, per variant, interior-point iterations, corrector solves skipped, and whether the exchange afterwards still
starts from as good a guess.

  python scripts/proto_ipm_skip.py [--n 2000] [--count 4]
"""
import argparse
import os
import sys

import numpy as np
import scipy.linalg as sla

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from proto_ipm import problem, identify, exact_active_set          # noqa: E402
from global_racetrajectory_optimization_amd import synthetic        # noqa: E402


def ipm_skip(H, f, lo, hi, thr=None, mode="affine", tol=1e-10, gamma=0.995, max_iter=60, log=False):
    n = len(f)
    x = 0.5 * (lo + hi)
    g = H @ x + f
    zscale = np.abs(g).max()
    wmean = (hi - lo).mean()
    zl = np.full(n, zscale)
    zu = np.full(n, zscale)
    skipped = 0
    for it in range(1, max_iter + 1):
        g = H @ x + f
        sl, su = x - lo, hi - x
        mu = (sl @ zl + su @ zu) / (2 * n)
        rd = np.abs(g - zl + zu).max()
        if mu < tol * zscale * wmean and rd < tol * zscale:
            return x, zl, zu, it - 1, skipped
        sig = zl / sl + zu / su
        cf = sla.cho_factor(H + np.diag(sig))
        dxa = sla.cho_solve(cf, -g)
        dzla = -zl - zl * dxa / sl
        dzua = -zu + zu * dxa / su

        def lengths(dx, dzl, dzu):
            ap = np.inf
            m = dx < 0
            if m.any(): ap = min(ap, (-sl[m] / dx[m]).min())
            m = dx > 0
            if m.any(): ap = min(ap, (su[m] / dx[m]).min())
            ad = np.inf
            m = dzl < 0
            if m.any(): ad = min(ad, (-zl[m] / dzl[m]).min())
            m = dzu < 0
            if m.any(): ad = min(ad, (-zu[m] / dzu[m]).min())
            return ap, ad
        ap, ad = lengths(dxa, dzla, dzua)
        ap1, ad1 = min(ap, 1.0), min(ad, 1.0)
        mua = ((sl + ap1 * dxa) @ (zl + ad1 * dzla) + (su - ap1 * dxa) @ (zu + ad1 * dzua)) / (2 * n)
        sigma = (mua / mu) ** 3
        skip = thr is not None and min(ap1, ad1) >= thr
        if skip and mode == "affine":
            a = min(1.0, gamma * min(ap, ad))
            dx, dzl, dzu = dxa, dzla, dzua
            skipped += 1
        elif skip and mode == "diag":
            # the corrector's right-hand side WITHOUT a second solve: the second-order and centring terms go into the multipliers only
            # (dz gets the terms, dx stays the affine one)
            smu = sigma * mu
            dx = dxa
            dzl = (-sl * zl + smu - dxa * dzla - zl * dx) / sl
            dzu = (-su * zu + smu + dxa * dzua + zu * dx) / su
            ap2, ad2 = lengths(dx, dzl, dzu)
            a = min(1.0, gamma * min(ap2, ad2))
            skipped += 1
        else:
            smu = sigma * mu
            rhs = -g + (smu - dxa * dzla) / sl - (smu + dxa * dzua) / su
            dx = sla.cho_solve(cf, rhs)
            dzl = (-sl * zl + smu - dxa * dzla - zl * dx) / sl
            dzu = (-su * zu + smu + dxa * dzua + zu * dx) / su
            ap2, ad2 = lengths(dx, dzl, dzu)
            a = min(1.0, gamma * min(ap2, ad2))
        if log:
            print("   it %2d mu %.2e rd %.2e ap %.3f ad %.3f sigma %.1e %s a %.4f" % (it, mu / (zscale * wmean), rd / zscale, ap1, ad1, sigma, "SKIP" if skip else "    ", a))
        x = x + a * dx
        zl = zl + a * dzl
        zu = zu + a * dzu
    return x, zl, zu, max_iter, skipped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--count", type=int, default=4)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--log", action="store_true")
    args = ap.parse_args()
    ref_b, nv_b, sc_b = synthetic.oval_batch(args.count, n=args.n, first=args.first)
    variants = [("mehrotra (kernel)", dict()), ("skip thr 0.95 affine", dict(thr=0.95)), ("skip thr 0.9 affine", dict(thr=0.9)),
                ("skip thr 0.8 affine", dict(thr=0.8)), ("skip thr 0.9 diag", dict(thr=0.9, mode="diag")), ("skip thr 0.8 diag", dict(thr=0.8, mode="diag")),
                ("skip thr 0.6 diag", dict(thr=0.6, mode="diag"))]
    tot = {name: [0, 0, 0] for name, _ in variants}
    for k in range(args.count):
        H, f, lo, hi = problem(ref_b[k], nv_b[k], sc_b[k])
        zscale = np.abs(H @ (0.5 * (lo + hi)) + f).max()
        print("problem %d" % k, flush=True)
        for name, kw in variants:
            x, zl, zu, it, sk = ipm_skip(H, f, lo, hi, log=args.log and k == 0, **kw)
            st = identify(x, zl, zu, lo, hi, zscale)
            xs, st2, asit = exact_active_set(H, f, lo, hi, st)
            # cost in units of (factorisation + fused solve) = 0.881 + 0.080 and own-chain solve 0.496 + second pass 0.13 (MB per 2000 waypoints)
            cost = it * (1.762 + 0.16 + 0.53) + (it - sk) * (0.992 + 0.26)
            tot[name][0] += it; tot[name][1] += sk; tot[name][2] += cost
            print("  %-24s ipm %2d  skipped %2d  as %2d  guess wrong %3d   ~MB %.1f" % (name, it, sk, asit, int((st != st2).sum()), cost), flush=True)
    print("totals over %d problems:" % args.count)
    for name, _ in variants:
        print("  %-24s ipm %.2f skipped %.2f  ~MB per problem %.2f" % (name, tot[name][0] / args.count, tot[name][1] / args.count, tot[name][2] / args.count))


if __name__ == "__main__":
    main()
