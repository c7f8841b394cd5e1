"""Prototype: Goldfarb-Idnani in curvature coordinates xi = E alpha (least-distance form), QR by Gram-Schmidt with
re-orthogonalisation, deletes by Givens.  Dense E here; the kernel takes E / E^-1 through the spline system."""
import sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import tph_ref, qp_ref

def gi_xi(E, kref, lo, hi, kb, use_kappa=True, maxit=100000, verbose=False):
    n = E.shape[0]
    Einv = np.linalg.inv(E)
    xi = -2.0 * kref
    al = Einv @ xi
    # working set
    W = []            # list of (kind, idx, sign): kind 0 box, 1 kappa; sign +1: lower (n'xi >= b), -1 upper
    Nw = np.zeros((n, 0)); R = np.zeros((0, 0)); u = np.zeros(0)
    inW = {}
    nadd = ndrop = 0
    wscale = np.mean(hi - lo)
    it = 0
    while True:
        # most violated
        vb_lo = (lo - al) / wscale; vb_hi = (al - hi) / wscale
        kap = kref + xi
        vk_hi = (kap - kb) / kb if use_kappa else np.full(n, -1.0); vk_lo = (-kb - kap) / kb if use_kappa else np.full(n, -1.0)
        cands = [(vb_lo, 0, +1), (vb_hi, 0, -1), (vk_lo, 1, +1), (vk_hi, 1, -1)]
        best = 1e-11; p = None
        for arr, kind, sg in cands:
            a2 = arr.copy()
            for (k2, i2, s2) in W:
                if k2 == kind and s2 == sg: a2[i2] = -1.0
            i = int(np.argmax(a2))
            if a2[i] > best: best = a2[i]; p = (kind, i, sg)
        if p is None: break
        kind, i, sg = p
        if kind == 0:
            npv = sg * Einv[i, :]        # E^-T e_i
        else:
            npv = np.zeros(n); npv[i] = sg
        up = 0.0
        while True:
            it += 1
            if it > maxit: return None
            q = Nw.shape[1]
            if q:
                w = Nw.T @ npv
                v = np.linalg.solve(R.T, w); r = np.linalg.solve(R, v)
                d2 = npv - Nw @ r
                for _ in range(2):
                    w2 = Nw.T @ d2; v2 = np.linalg.solve(R.T, w2); r2 = np.linalg.solve(R, v2)
                    d2 = d2 - Nw @ r2; r += r2; v += v2
            else:
                v = np.zeros(0); r = np.zeros(0); d2 = npv.copy()
            rho2 = d2 @ d2
            dep = rho2 <= (1e-13 ** 2) * (npv @ npv)
            # slack of p
            if kind == 0: sp = sg * al[i] - (lo[i] if sg > 0 else -hi[i])
            else: sp = sg * (kref[i] + xi[i]) - (-kb)
            t1 = np.inf; l = -1
            for k in range(q):
                if r[k] > 0 and u[k] / r[k] < t1: t1 = u[k] / r[k]; l = k
            t2 = np.inf if dep else -sp / rho2
            t = min(t1, t2)
            if np.isinf(t): return "infeasible"
            if not np.isinf(t2):
                xi = xi + t * d2
                al = al + t * (Einv @ d2)
            u = u - t * r; up += t
            if not np.isinf(t2) and t2 <= t1:
                Rn = np.zeros((q + 1, q + 1)); Rn[:q, :q] = R; Rn[:q, q] = v; Rn[q, q] = np.sqrt(rho2)
                R = Rn; Nw = np.column_stack((Nw, npv)); u = np.append(u, up); W.append(p); nadd += 1
                break
            # drop l
            W.pop(l); u = np.delete(u, l); Nw = np.delete(Nw, l, axis=1)
            R = np.delete(R, l, axis=1)
            for j in range(l, q - 1):
                a, b = R[j, j], R[j + 1, j]; h = np.hypot(a, b)
                if h == 0: continue
                c, s = a / h, b / h
                rj = c * R[j, :] + s * R[j + 1, :]; rj1 = -s * R[j, :] + c * R[j + 1, :]
                R[j, :] = rj; R[j + 1, :] = rj1
            R = R[:q - 1, :]
            ndrop += 1
    return al, W, u, (nadd, ndrop, it)

def run(ref, nv, A, kb, wveh, name):
    H, f, E, kref, aux = tph_ref.assemble_dense(ref, nv, A)
    n = ref.shape[0]
    hi = ref[:, 2] - wveh / 2; lo = -(ref[:, 3] - wveh / 2)
    info = {}
    G, h = tph_ref.constraints_dense(ref, E, kref, kb, wveh)
    t0 = time.time(); a_ref = qp_ref.solve_qp_gi(H, f, G, h, info); t1 = time.time()
    out = gi_xi(E, kref, lo, hi, kb)
    t2 = time.time()
    if out is None or isinstance(out, str): print(name, "FAILED", out); return
    al, W, u, cnt = out
    print("%s n=%d: dense GI iters %s (%.1fs); xi-GI add/drop/it %s (%.1fs); |W|=%d (kappa %d); max|da|=%.2e min u=%.2e" % (
        name, n, info["iters"], t1 - t0, cnt, t2 - t1, len(W), sum(1 for w in W if w[0] == 1), np.max(np.abs(al - a_ref)), u.min() if len(u) else 0))

if __name__ == "__main__":
    from test_emu_kernels import stadium_problem
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which in ("small", "all"):
        for t in ("rounded_rectangle", "handling_track"):
            g = np.load("" + ROOT + "/tests/golden/%s.npz" % t)
            n = g["reftrack"].shape[0]
            _, _, A, _ = tph_ref.calc_splines(np.vstack((g["reftrack"][:, :2], g["reftrack"][:1, :2])))
            run(g["reftrack"], g["normvec"], A, 0.12, 3.4, t)
            run(g["reftrack"], g["normvec"], A, 0.07, 3.4, t + " kb 0.07")
    if which in ("stadium", "all"):
        for n in (360, 720):
            ref, nv, A, sc, kb = stadium_problem(n, 0.0223)
            run(ref, nv, A, kb, 2.0, "stadium%d" % n)
    if which in ("berlin", "all"):
        g = np.load("" + ROOT + "/tests/golden/berlin_2018.npz")
        _, _, A, _ = tph_ref.calc_splines(np.vstack((g["reftrack"][:, :2], g["reftrack"][:1, :2])))
        run(g["reftrack"], g["normvec"], A, 0.12, 3.4, "berlin")
        run(g["reftrack"], g["normvec"], A, 0.07, 3.4, "berlin kb 0.07")
