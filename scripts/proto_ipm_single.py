"""Prototype: one solve per interior-point iteration (centring parameter and second-order term LAGGED from the previous iteration) against Mehrotra's two."""
import sys, numpy as np, scipy.linalg as sla
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from proto_ipm import problem, identify, exact_active_set
from global_racetrajectory_optimization_amd import synthetic

def steps(dx, dzl, dzu, sl, su, zl, zu):
    a = np.inf
    for d, v in ((dx, -sl / np.where(dx < 0, dx, -1e-300)), ):
        pass
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

def ipm_single(H, f, lo, hi, tol=1e-10, mode="lag", max_iter=80, log=False):
    n = len(f)
    x = 0.5 * (lo + hi); g = H @ x + f
    zscale = np.abs(g).max(); wmean = (hi - lo).mean()
    zl = np.full(n, zscale); zu = np.full(n, zscale)
    sigma = 0.2; a_prev = 1.0
    corr_l = np.zeros(n); corr_u = np.zeros(n)
    for it in range(1, max_iter + 1):
        g = H @ x + f
        sl, su = x - lo, hi - x
        mu = (sl @ zl + su @ zu) / (2 * n)
        rd = np.abs(g - zl + zu).max()
        if mu < tol * zscale * wmean and rd < tol * zscale:
            return it - 1
        sig = zl / sl + zu / su
        cf = sla.cho_factor(H + np.diag(sig))
        smu = sigma * mu
        tl = -sl * zl + smu - corr_l
        tu = -su * zu + smu - corr_u
        rhs = -(g - zl + zu) + tl / sl - tu / su
        dx = sla.cho_solve(cf, rhs)
        dzl = (tl - zl * dx) / sl; dzu = (tu + zu * dx) / su
        gm = min(max(0.995, 1.0 - 10.0 * mu / (zscale * wmean)), 1.0 - 1e-9)
        ap, ad = steps(dx, dzl, dzu, sl, su, zl, zu)
        a = min(1.0, gm * min(ap, ad))
        # predicted complementarity after the step -> next sigma (Mehrotra's cube rule on the realised reduction)
        mun = ((sl + a * dx) @ (zl + a * dzl) + (su - a * dx) @ (zu + a * dzu)) / (2 * n)
        x = x + a * dx; zl = zl + a * dzl; zu = zu + a * dzu
        if mode == "lag":
            sigma = min(max((mun / mu) ** 2, 1e-3), 0.5) if a > 0.9 else min(0.5, max((1 - a) ** 2, 0.05))
            corr_l = 0 * corr_l; corr_u = 0 * corr_u
        elif mode == "lag2":       # second-order term from this step, used in the next (scaled by the step actually taken)
            sigma = min(max((mun / mu) ** 2, 1e-3), 0.5) if a > 0.9 else min(0.5, max((1 - a) ** 2, 0.05))
            corr_l = (1 - a) * (dx * dzl); corr_u = -(1 - a) * (dx * dzu)
        if log: print("   it %2d mu %.2e rd %.2e a %.4f sigma_next %.3f" % (it, mu / (zscale * wmean), rd / zscale, a, sigma))
    return max_iter

if __name__ == "__main__":
    from proto_ipm import ipm as ipm_mehrotra
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    ref_b, nv_b, sc_b = synthetic.oval_batch(3, n=n, first=0)
    for k in range(3):
        H, f, lo, hi = problem(ref_b[k], nv_b[k], sc_b[k])
        x, zl, zu, it, hist = ipm_mehrotra(H, f, lo, hi, adaptive=True)
        print("problem %d: mehrotra %d iterations (cost %.1f); single-solve lag %d (cost %.1f), lag2 %d" % (
            k, it, it * 1.0, ipm_single(H, f, lo, hi, mode="lag", log=(k == 0)), ipm_single(H, f, lo, hi, mode="lag") * 0.73, ipm_single(H, f, lo, hi, mode="lag2")))
