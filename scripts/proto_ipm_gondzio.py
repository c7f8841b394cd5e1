"""Prototype: Gondzio multiple centrality correctors / step rules on the kernel's Mehrotra iteration (dense H), N = 2000 ovals."""
import sys, time
import numpy as np, scipy.linalg as sla
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from proto_ipm import problem, identify, exact_active_set
from global_racetrajectory_optimization_amd import synthetic

def ipm(H, f, lo, hi, tol=1e-10, kcorr=0, corr_when=0.9, beta_min=0.1, beta_max=10.0, delta=0.3, split=False, max_iter=60, log=False):
    n = len(f)
    x = 0.5 * (lo + hi); g = H @ x + f
    zscale = np.abs(g).max(); wmean = (hi - lo).mean()
    zl = np.full(n, zscale); zu = np.full(n, zscale)
    nsolve = 0
    for it in range(1, max_iter + 1):
        g = H @ x + f
        sl, su = x - lo, hi - x
        mu = (sl @ zl + su @ zu) / (2 * n)
        rd = np.abs(g - zl + zu).max()
        if mu < tol * zscale * wmean and rd < tol * zscale:
            return x, zl, zu, it - 1, nsolve
        sig = zl / sl + zu / su
        cf = sla.cho_factor(H + np.diag(sig))
        dxa = sla.cho_solve(cf, -g); nsolve += 1
        dzla = -zl - zl * dxa / sl
        dzua = -zu + zu * dxa / su
        def steps(dx, dzl, dzu):
            ap = np.inf; ad = np.inf
            m = dx < 0
            if m.any(): ap = min(ap, (-sl[m] / dx[m]).min())
            m = dx > 0
            if m.any(): ap = min(ap, (su[m] / dx[m]).min())
            m = dzl < 0
            if m.any(): ad = min(ad, (-zl[m] / dzl[m]).min())
            m = dzu < 0
            if m.any(): ad = min(ad, (-zu[m] / dzu[m]).min())
            return ap, ad
        ap, ad = steps(dxa, dzla, dzua); ap = min(ap, 1.0); ad = min(ad, 1.0)
        mua = ((sl + ap * dxa) @ (zl + ad * dzla) + (su - ap * dxa) @ (zu + ad * dzua)) / (2 * n)
        smu = (mua / mu) ** 3 * mu
        # right-hand sides in terms of complementarity targets: sl*dzl + zl*dx = tl, su*dzu - zu*dx = tu
        tl = -sl * zl + smu - dxa * dzla
        tu = -su * zu + smu + dxa * dzua
        def solve_for(tl, tu):
            rhs = -g + (tl + sl * zl) / sl - (tu + su * zu) / su      # = -g + zl + tl/sl ... careful below
            return rhs
        # reduced system: (H + sig) dx = -(g - zl + zu) + ... derive: dzl = (tl - zl dx)/sl, dzu = (tu + zu dx)/su
        # stationarity: H dx - dzl + dzu = -(g - zl + zu)  ->  (H + sig) dx = -(g - zl + zu) + tl/sl - tu/su
        def direction(tl, tu):
            rhs = -(g - zl + zu) + tl / sl - tu / su
            dx = sla.cho_solve(cf, rhs)
            return dx, (tl - zl * dx) / sl, (tu + zu * dx) / su
        dx, dzl, dzu = direction(tl, tu); nsolve += 1
        gm = min(max(0.995, 1.0 - 10.0 * mu / (zscale * wmean)), 1.0 - 1e-9)
        apx, adx = steps(dx, dzl, dzu)
        a = min(1.0, gm * min(apx, adx))
        k = 0
        while k < kcorr and a < corr_when:
            # Gondzio: aim for a longer step, project the complementarity products of the trial point into [beta_min, beta_max] * smu
            at = min(1.0, a + delta)
            vl = (sl + at * dx) * (zl + at * dzl); vu = (su - at * dx) * (zu + at * dzu)
            def proj(v):
                t = np.zeros_like(v)
                t[v < beta_min * smu] = (beta_min * smu - v)[v < beta_min * smu]
                t[v > beta_max * smu] = (beta_max * smu - v)[v > beta_max * smu]
                t[t < -beta_max * smu] = -beta_max * smu
                return t
            tl2 = tl + proj(vl); tu2 = tu + proj(vu)
            dx2, dzl2, dzu2 = direction(tl2, tu2); nsolve += 1
            ap2, ad2 = steps(dx2, dzl2, dzu2)
            a2 = min(1.0, gm * min(ap2, ad2))
            if a2 >= a + 0.1 * delta:
                dx, dzl, dzu, a, tl, tu = dx2, dzl2, dzu2, a2, tl2, tu2
                k += 1
            else:
                break
        if split:
            aP = min(1.0, gm * apx if k == 0 else gm * steps(dx, dzl, dzu)[0]); aD = min(1.0, gm * (adx if k == 0 else steps(dx, dzl, dzu)[1]))
            x = x + aP * dx; zl = zl + aD * dzl; zu = zu + aD * dzu
        else:
            x = x + a * dx; zl = zl + a * dzl; zu = zu + a * dzu
        if log: print("   it %2d mu %.2e rd %.2e a %.4f corr %d" % (it, mu / (zscale * wmean), rd / zscale, a, k))
    return x, zl, zu, max_iter, nsolve

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ref_b, nv_b, sc_b = synthetic.oval_batch(cnt, n=n, first=0)
    variants = {"kernel": dict(), "gondzio k1": dict(kcorr=1), "gondzio k2": dict(kcorr=2), "gondzio k1 always": dict(kcorr=1, corr_when=1.01),
                "gondzio k2 d0.5": dict(kcorr=2, delta=0.5), "split": dict(split=True)}
    tot = {k: [0, 0, 0] for k in variants}
    for k in range(cnt):
        H, f, lo, hi = problem(ref_b[k], nv_b[k], sc_b[k])
        zscale = np.abs(H @ (0.5 * (lo + hi)) + f).max()
        for name, kw in variants.items():
            x, zl, zu, it, ns = ipm(H, f, lo, hi, log=(k == 0 and name in ("kernel", "gondzio k2")), **kw)
            st = identify(x, zl, zu, lo, hi, zscale)
            xs, st2, asit = exact_active_set(H, f, lo, hi, st)
            tot[name][0] += it; tot[name][1] += ns; tot[name][2] += asit
            print("problem %d %-20s ipm %2d solves %2d (cost %.2f iteration-equivalents) as %d wrong %d" % (k, name, it, ns, it + 0.22 * (ns - 2 * it), asit, int((st != st2).sum())))
    for name, (it, ns, a) in tot.items():
        print("%-20s mean ipm %.2f, extra solves %.2f, cost %.2f, as %.2f" % (name, it / cnt, (ns - 2 * it) / cnt, (it + 0.22 * (ns - 2 * it)) / cnt, a / cnt))
