"""
Design prototype (numpy) of the structure-exploiting formulation the HIP kernels implement -- kept in the repo as the
executable derivation behind DESIGN.md section 3/4.  Not product code, not the oracle.

  band_assembly   : E_kappa (cyclic band), k_ref, x', y' from [x,y], normals, spline scalings
  bpp_box         : block-principal-pivoting active-set on the box-constrained QP (dense solves here)
"""
import sys
import numpy as np


def periodic_pivots(sub, diag, sup, warm=48):
    n = diag.size
    de = np.empty(n)
    ep = np.empty(n)
    d = diag[(-warm) % n]
    for k in range(-warm + 1, n):
        m = k % n
        d = diag[m] - sub[m] * sup[(m - 1) % n] / d
        if k >= 0:
            de[m] = d
    e = diag[(n - 1 + warm) % n]
    for k in range(n - 1 + warm - 1, -1, -1):
        m = k % n
        e = diag[m] - sup[m] * sub[(m + 1) % n] / e
        if k < n:
            ep[m] = e
    return de, ep


def tinv_rows(sub, diag, sup, W):
    """g[i, W + k] = (T_cyclic^-1)[i, (i + k) mod n] accumulated from the periodic Green's function, |k| <= W."""
    n = diag.size
    de, ep = periodic_pivots(sub, diag, sup)
    g = np.zeros((n, 2 * W + 1))
    idx = np.arange(n)
    g0 = 1.0 / (de + ep - diag)
    g[:, W] = g0
    cur = g0.copy()
    for k in range(0, W):          # right: G[i, j+1] = -(sup[j] / ep[j+1]) G[i, j],  j = i + k
        j = (idx + k) % n
        cur = -(sup[j] / ep[(j + 1) % n]) * cur
        g[:, W + k + 1] = cur
    cur = g0.copy()
    for k in range(0, W):          # left: G[i, j-1] = -(sub[j] / de[j-1]) G[i, j],  j = i - k
        j = (idx - k) % n
        cur = -(sub[j] / de[(j - 1) % n]) * cur
        g[:, W - k - 1] = cur
    return g


def band_assembly(xy, nv, s, bE=32):
    """Returns Eb [n, 2bE+1] with Eb[i, bE+k] = E[i, (i+k) mod n], k_ref, xp, yp, (Db, cx, cy)."""
    n = xy.shape[0]
    W = bE + 2
    s_prev = np.roll(s, 1)          # s_{m-1}
    sub = np.ones(n)
    diag = 2.0 * s_prev ** 2 + 2.0 * s_prev
    sup = s_prev * s ** 2
    g = tinv_rows(sub, diag, sup, W)
    idx = np.arange(n)
    delta = np.roll(xy, -1, axis=0) - xy                                   # D_i = p_{i+1} - p_i
    rhs = 3.0 * (s_prev[:, None] * delta - np.roll(delta, 1, axis=0))       # centre m: 3 (s_{m-1} D_m - D_{m-1})
    c = np.zeros((n, 2))
    for k in range(-W, W + 1):
        c += g[:, W + k, None] * rhs[(idx + k) % n]
    # Cc[i, j] = 3 [ G[i, j+1] - (1 + s_{j-1}) G[i, j] + s_{j-2} G[i, j-1] ],  D = 2 Cc
    Db = np.zeros((n, 2 * bE + 1))
    for k in range(-bE, bE + 1):
        j = (idx + k) % n
        Db[:, bE + k] = 6.0 * (g[:, W + k + 1] - (1.0 + s[(j - 1) % n]) * g[:, W + k] + s[(j - 2) % n] * g[:, W + k - 1])
    c_next = np.roll(c, -1, axis=0)
    b = delta - (2.0 * c + (s ** 2)[:, None] * c_next) / 3.0
    xp, yp = b[:, 0], b[:, 1]
    xpp, ypp = 2.0 * c[:, 0], 2.0 * c[:, 1]
    den = (xp ** 2 + yp ** 2) ** 1.5
    cp = 1.0 / den
    k_ref = cp * (xp * ypp - yp * xpp)
    Eb = np.zeros_like(Db)
    for k in range(-bE, bE + 1):
        j = (idx + k) % n
        Eb[:, bE + k] = Db[:, bE + k] * cp * (xp * nv[j, 1] - yp * nv[j, 0])
    return Eb, k_ref, xp, yp, Db, c


def band_to_dense(Eb):
    n, w = Eb.shape
    b = (w - 1) // 2
    E = np.zeros((n, n))
    idx = np.arange(n)
    for k in range(-b, b + 1):
        E[idx, (idx + k) % n] += Eb[:, b + k]
    return E


def bpp_box(H, f, lo, hi, max_iter=200, p_max=3, verbose=False):
    """Kim-Park / Judice-Pires block principal pivoting for min 1/2 x'Hx + f'x, lo <= x <= hi."""
    n = f.size
    state = np.zeros(n, dtype=int)       # 0 free, -1 at lo, +1 at hi
    p = p_max
    best = n + 1
    x = np.zeros(n)
    for it in range(1, max_iter + 1):
        F = state == 0
        x = np.where(state < 0, lo, np.where(state > 0, hi, 0.0))
        if F.any():
            rhs = -(f[F] + H[np.ix_(F, ~F)] @ x[~F])
            x[F] = np.linalg.solve(H[np.ix_(F, F)], rhs)
        y = H @ x + f                    # gradient; need y >= 0 at lo, y <= 0 at hi
        tol = 1e-12
        v_lo = F & (x < lo - tol)
        v_hi = F & (x > hi + tol)
        v_rl = (state < 0) & (y < -0)    # at lower bound but gradient wants to increase... y<0 => release
        v_ru = (state > 0) & (y > 0)
        V = v_lo | v_hi | v_rl | v_ru
        nv = int(V.sum())
        if verbose:
            print("  it", it, "infeas", nv, "free", int(F.sum()))
        if nv == 0:
            return x, state, it
        if nv < best:
            best = nv
            p = p_max
            full = True
        elif p > 0:
            p -= 1
            full = True
        else:
            full = False
        if full:
            state[v_lo] = -1
            state[v_hi] = 1
            state[v_rl | v_ru] = 0
        else:
            i = int(np.max(np.where(V)[0]))
            if v_lo[i]:
                state[i] = -1
            elif v_hi[i]:
                state[i] = 1
            else:
                state[i] = 0
    return x, state, max_iter


if __name__ == "__main__":
    sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/global_racetrajectory_optimization_amd')
    import trajectory_planning_helpers as tph
    for name in ["rounded_rectangle", "handling_track", "modena_2019", "berlin_2018"]:
        reg = np.load(f"/tmp/reftrack_{name}.npy")
        dz = np.load(f"/tmp/dense_{name}.npz")
        path_cl = np.vstack((reg[:, :2], reg[0, :2]))
        s = tph.calc_splines.spline_scalings(path_cl)
        for bE in (24, 32, 40):
            if 2 * bE + 1 > reg.shape[0]:
                continue
            Eb, k_ref, xp, yp, Db, c = band_assembly(reg[:, :2], dz["nv"], s, bE)
            E = band_to_dense(Eb)
            print(name, "bE", bE, "E err", np.abs(E - dz["E"]).max() / np.abs(dz["E"]).max(), "kref err", np.abs(k_ref - dz["k_ref"]).max())
            H = E.T @ E
            f = 2 * E.T @ k_ref
            lo = -(reg[:, 3] - 1.7); hi = reg[:, 2] - 1.7
            x, st, it = bpp_box(H, f, lo, hi)
            print("    bpp iters", it, "nact", int((st != 0).sum()), "alpha diff vs dense GI", np.abs(x - dz["alpha"]).max())
