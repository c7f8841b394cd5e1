"""
Design prototype (numpy): the Newton system of the interior point,  (H + diag(sig)) dx = r  with  H = E' diag(w) E,  solved WITHOUT forming
H -- through the sparse saddle-point system in (alpha, cx, cy, lx, ly) per waypoint, which is cyclic block-tridiagonal with 5 x 5 blocks:

    kappa_i = k_ref_i + a_i cx_i + b_i cy_i,      T cx = R (px + Nx alpha),   T cy = R (py + Ny alpha)        (T, R cyclic tridiagonal)

    alpha_m :  sig_m alpha_m - sum_i R[i, m] (nx_m lx_i + ny_m ly_i)           = r_m
    cx_m    :  w_m a_m (a_m cx_m + b_m cy_m) + sum_i T[i, m] lx_i               = 0
    cy_m    :  w_m b_m (a_m cx_m + b_m cy_m) + sum_i T[i, m] ly_i               = 0
    lx_m    :  sum_k T[m, k] cx_k - sum_k R[m, k] nx_k alpha_k                  = 0
    ly_m    :  sum_k T[m, k] cy_k - sum_k R[m, k] ny_k alpha_k                  = 0

Eliminating (c, l) gives back (sig + E' w E) alpha = r exactly (E = a T^-1 R Nx + b T^-1 R Ny is the UNtruncated band the kernels keep 65
diagonals of).  Question answered here: is a block-tridiagonal LU with pivoting inside the 5 x 5 diagonal blocks only -- segments between
separator waypoints eliminated independently, then a small reduced system on the separators -- accurate enough at the conditioning of a late
interior-point iteration (sig from 1e-8 to 1e12, pinned rows)?  Not product code, not the oracle.
"""
import sys
import numpy as np

sys.path.insert(0, "/root/repo/scripts")
from proto_banded import band_assembly, band_to_dense  # noqa: E402


def kkt_coeffs(xy, nv, s):
    n = len(s)
    s_prev = np.roll(s, 1)
    T = dict(sub=np.ones(n), diag=2.0 * s_prev ** 2 + 2.0 * s_prev, sup=s_prev * s ** 2)       # T[m, m-1], T[m, m], T[m, m+1]
    R = dict(sub=3.0 * np.ones(n), diag=-3.0 * (s_prev + 1.0), sup=3.0 * s_prev)                # R[m, m-1], R[m, m], R[m, m+1]
    Eb, k_ref, xp, yp, Db, c = band_assembly(xy, nv, s, 32)
    cp = 1.0 / (xp ** 2 + yp ** 2) ** 1.5
    a = -2.0 * cp * yp
    b = 2.0 * cp * xp
    return T, R, a, b, Eb, k_ref


def tri_dense(t, n):
    M = np.zeros((n, n))
    i = np.arange(n)
    M[i, (i - 1) % n] += t["sub"]
    M[i, i] += t["diag"]
    M[i, (i + 1) % n] += t["sup"]
    return M


def kkt_blocks(T, R, a, b, nv, sig, w, pinned):
    """Cyclic block-tridiagonal: Lo[m] couples point m to m-1, Dg[m] to itself, Up[m] to m+1.  Order (alpha, cx, cy, lx, ly)."""
    n = len(a)
    Lo = np.zeros((n, 5, 5)); Dg = np.zeros((n, 5, 5)); Up = np.zeros((n, 5, 5))
    nx, ny = nv[:, 0], nv[:, 1]
    for m in range(n):
        mp, mn = (m - 1) % n, (m + 1) % n
        # alpha_m row: sig alpha_m - sum_i R[i,m] (nx_m lx_i + ny_m ly_i);  R[m-1, m] = R.sup[m-1], R[m, m] = R.diag[m], R[m+1, m] = R.sub[m+1]
        if pinned[m]:
            Dg[m, 0, 0] = 1.0
        else:
            Dg[m, 0, 0] = sig[m]
            Dg[m, 0, 3] = -R["diag"][m] * nx[m]; Dg[m, 0, 4] = -R["diag"][m] * ny[m]
            Lo[m, 0, 3] = -R["sup"][mp] * nx[m]; Lo[m, 0, 4] = -R["sup"][mp] * ny[m]
            Up[m, 0, 3] = -R["sub"][mn] * nx[m]; Up[m, 0, 4] = -R["sub"][mn] * ny[m]
        # cx_m, cy_m rows
        Dg[m, 1, 1] = w[m] * a[m] * a[m]; Dg[m, 1, 2] = w[m] * a[m] * b[m]
        Dg[m, 2, 1] = w[m] * a[m] * b[m]; Dg[m, 2, 2] = w[m] * b[m] * b[m]
        Dg[m, 1, 3] = T["diag"][m]; Dg[m, 2, 4] = T["diag"][m]
        Lo[m, 1, 3] = T["sup"][mp]; Lo[m, 2, 4] = T["sup"][mp]          # T[m-1, m]
        Up[m, 1, 3] = T["sub"][mn]; Up[m, 2, 4] = T["sub"][mn]          # T[m+1, m]
        # lx_m, ly_m rows: T[m, k] c_k - R[m, k] n_k alpha_k   (a pinned alpha_k is zero: its column is dropped)
        Dg[m, 3, 1] = T["diag"][m]; Dg[m, 4, 2] = T["diag"][m]
        Lo[m, 3, 1] = T["sub"][m]; Lo[m, 4, 2] = T["sub"][m]
        Up[m, 3, 1] = T["sup"][m]; Up[m, 4, 2] = T["sup"][m]
        if not pinned[m]:
            Dg[m, 3, 0] = -R["diag"][m] * nx[m]; Dg[m, 4, 0] = -R["diag"][m] * ny[m]
        if not pinned[mp]:
            Lo[m, 3, 0] = -R["sub"][m] * nx[mp]; Lo[m, 4, 0] = -R["sub"][m] * ny[mp]
        if not pinned[mn]:
            Up[m, 3, 0] = -R["sup"][m] * nx[mn]; Up[m, 4, 0] = -R["sup"][m] * ny[mn]
    return Lo, Dg, Up


def lu5(A):
    """5 x 5 LU with partial pivoting, returns a solver for right-hand sides [5, k]."""
    A = A.copy()
    piv = np.arange(5)
    for k in range(5):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            piv[[k, p]] = piv[[p, k]]
        A[k + 1:, k] /= A[k, k]
        A[k + 1:, k + 1:] -= np.outer(A[k + 1:, k], A[k, k + 1:])

    def solve(Bm):
        Y = Bm[piv].astype(np.float64).copy()
        for k in range(5):
            Y[k + 1:] -= np.outer(A[k + 1:, k], Y[k])
        for k in range(4, -1, -1):
            Y[k] /= A[k, k]
            Y[:k] -= np.outer(A[:k, k], Y[k])
        return Y
    return solve, np.abs(np.diag(A)).min(), np.abs(A).max()


def segment_solve(Lo, Dg, Up, rhs_cols, lo, hi):
    """Interior points lo .. hi-1 (non-cyclic block tridiagonal): block LU forward, back substitution, for the columns rhs_cols [len, 5, k]."""
    L = hi - lo
    S = [None] * L
    Y = np.zeros_like(rhs_cols)
    G = [None] * L          # G_k = S_k^-1 Up_k
    stats = []
    Dk = Dg[lo].copy()
    Yk = rhs_cols[0].copy()
    for k in range(L):
        m = lo + k
        if k > 0:
            Dk = Dg[m] - Lo[m] @ G[k - 1]
            Yk = rhs_cols[k] - Lo[m] @ Y[k - 1]
        solve, pmin, amax = lu5(Dk)
        stats.append((pmin, amax))
        S[k] = solve
        G[k] = solve(Up[m])
        Y[k] = solve(Yk)
    X = np.zeros_like(rhs_cols)
    X[L - 1] = Y[L - 1]
    for k in range(L - 2, -1, -1):
        X[k] = Y[k] - G[k] @ X[k + 1]
    return X, stats


def kkt_solve(Lo, Dg, Up, rhs, nseg):
    """Separators at points sep[j]; the segments between them are eliminated independently with 11 columns (rhs + coupling to both separators)."""
    n = Dg.shape[0]
    sep = [(j * n) // nseg for j in range(nseg)]
    x = np.zeros((n, 5))
    red = np.zeros((5 * nseg, 5 * nseg))
    rr = np.zeros(5 * nseg)
    seg_sol = []
    allstats = []
    for j in range(nseg):
        s0, s1 = sep[j], sep[(j + 1) % nseg] if j + 1 < nseg else sep[0] + n
        lo, hi = s0 + 1, s1
        L = hi - lo
        idx = [(lo + k) % n for k in range(L)]
        cols = np.zeros((L, 5, 11))
        cols[:, :, 0] = rhs[idx]
        cols[0, :, 1:6] = Lo[idx[0]]            # coupling of the first interior point to separator j (moved to the right-hand side with a minus later)
        cols[L - 1, :, 6:11] += Up[idx[-1]]     # coupling of the last interior point to separator j+1
        X, st = segment_solve(Lo[idx], Dg[idx], Up[idx], cols, 0, L)
        allstats += st
        seg_sol.append((idx, X))
        # separator j's row: Dg[s0] x_s0 + Up[s0] x_first + Lo[s0] x_(last of previous segment) = rhs[s0]
        # x_first = X[0][:,0] - X[0][:,1:6] x_sepj - X[0][:,6:11] x_sepj+1
        jn = (j + 1) % nseg
        r0 = 5 * j
        red[r0:r0 + 5, 5 * j:5 * j + 5] += Dg[s0 % n] - Up[s0 % n] @ X[0][:, 1:6]
        red[r0:r0 + 5, 5 * jn:5 * jn + 5] += -Up[s0 % n] @ X[0][:, 6:11]
        rr[r0:r0 + 5] += rhs[s0 % n] - Up[s0 % n] @ X[0][:, 0]
        # separator j+1's row gets Lo[s1] x_last
        r1 = 5 * jn
        red[r1:r1 + 5, 5 * j:5 * j + 5] += -Lo[s1 % n] @ X[L - 1][:, 1:6]
        red[r1:r1 + 5, 5 * jn:5 * jn + 5] += -Lo[s1 % n] @ X[L - 1][:, 6:11]
        rr[r1:r1 + 5] += -Lo[s1 % n] @ X[L - 1][:, 0]
    xs = np.linalg.solve(red, rr).reshape(nseg, 5)
    for j in range(nseg):
        idx, X = seg_sol[j]
        jn = (j + 1) % nseg
        x[sep[j]] = xs[j]
        for k, m in enumerate(idx):
            x[m] = X[k][:, 0] - X[k][:, 1:6] @ xs[j] - X[k][:, 6:11] @ xs[jn]
    return x, allstats, np.linalg.cond(red)


def banded_chol_solve(M, r):
    Lc = np.linalg.cholesky(M)
    y = np.linalg.solve(Lc, r)
    return np.linalg.solve(Lc.T, y)


def longdouble_solve(M, r):
    A = np.concatenate((M.astype(np.longdouble), r.astype(np.longdouble)[:, None]), axis=1)
    n = M.shape[0]
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
        A[k + 1:, k:] -= np.outer(A[k + 1:, k] / A[k, k], A[k, k:])
    x = np.zeros(n, dtype=np.longdouble)
    for k in range(n - 1, -1, -1):
        x[k] = (A[k, n] - A[k, k + 1:n] @ x[k + 1:]) / A[k, k]
    return x


def run(name, seed=0, nseg=16, sig_lo=-8, sig_hi=12, pin_frac=0.15, kappa_w=False):
    z = np.load(f"/root/repo/tests/golden/{name}.npz")
    ref, nv, s = z["reftrack"], z["normvec"], z["scaling"]
    n = ref.shape[0]
    rng = np.random.default_rng(seed)
    T, R, a, b, Eb, k_ref = kkt_coeffs(ref[:, :2], nv, s)
    E = band_to_dense(Eb)
    Td, Rd = tri_dense(T, n), tri_dense(R, n)
    Es = np.diag(a) @ np.linalg.solve(Td, Rd @ np.diag(nv[:, 0])) + np.diag(b) @ np.linalg.solve(Td, Rd @ np.diag(nv[:, 1]))
    print(f"{name}: n = {n}, |E_struct - E_band| / |E| = {np.abs(Es - E).max() / np.abs(E).max():.2e}, cond(E'E) = {np.linalg.cond(E.T @ E):.2e}")
    sig = 10.0 ** rng.uniform(sig_lo, sig_hi, n)
    pinned = rng.uniform(size=n) < pin_frac
    w = 1.0 + (10.0 ** rng.uniform(-3, 10, n) if kappa_w else 0.0) * np.ones(n)
    r = rng.standard_normal(n)
    r[pinned] = 0.0
    M = Es.T @ (w[:, None] * Es) + np.diag(sig)
    M[pinned, :] = 0.0; M[:, pinned] = 0.0; M[pinned, pinned] = 1.0
    x_ref = np.asarray(longdouble_solve(M, r), dtype=np.float64) if n <= 700 else None
    x_ch = banded_chol_solve(M, r)
    Lo, Dg, Up = kkt_blocks(T, R, a, b, nv, sig, w, pinned)
    rhs = np.zeros((n, 5)); rhs[:, 0] = r
    x5, stats, cred = kkt_solve(Lo, Dg, Up, rhs, nseg)
    x_k = x5[:, 0]
    st = np.array(stats)
    scale = np.abs(x_ch).max()
    print(f"   cond(M) = {np.linalg.cond(M):.2e}; reduced system cond {cred:.2e}; block pivots min {st[:, 0].min():.2e}, max entry {st[:, 1].max():.2e}")
    res = lambda x: np.abs(M @ x - r).max() / (np.abs(M) @ np.abs(x) + np.abs(r)).max()
    print(f"   backward error: cholesky {res(x_ch):.2e}, kkt {res(x_k):.2e}")
    if x_ref is not None:
        print(f"   forward error vs long double: cholesky {np.abs(x_ch - x_ref).max() / scale:.2e}, kkt {np.abs(x_k - x_ref).max() / scale:.2e}")
    else:
        print(f"   kkt vs cholesky: {np.abs(x_k - x_ch).max() / scale:.2e}")


if __name__ == "__main__" and len(sys.argv) == 1:
    for nm in ("rounded_rectangle", "handling_track", "berlin_2018_n333"):
        for seed in (0, 1):
            run(nm, seed)
        run(nm, 2, kappa_w=True)
        run(nm, 3, sig_lo=-12, sig_hi=-6, pin_frac=0.0)
        run(nm, 4, nseg=4)


# ----------------------------------------------------------------------------------------------------------------------
# The algorithm as the kernel runs it: in-place Gauss-Jordan with partial pivoting inside the 5 x 5 diagonal block (columns 0..4 end up
# holding (P D)^-1, applied later to P t), stored factors, forward / backward passes per segment, and the separators' block-cyclic system
# eliminated by the same routine (one "segment" of nsep - 1 blocks with both spikes pointing at separator 0).
# ----------------------------------------------------------------------------------------------------------------------
def gj5(Mx):
    """Mx [5, 5 + k]: in-place Gauss-Jordan on the first five columns with row pivoting.  Returns (Mx, perm): Mx[:, :5] = (P A)^-1,
    Mx[:, 5:] = A^-1 B; perm = the row picked at each of the five stages."""
    Mx = Mx.copy()
    perm = np.zeros(5, dtype=int)
    for k in range(5):
        p = k + int(np.argmax(np.abs(Mx[k:, k])))
        perm[k] = p
        if p != k:
            Mx[[k, p]] = Mx[[p, k]]
        inv = 1.0 / Mx[k, k]
        mult = -Mx[:, k] * inv
        colk = mult.copy()
        colk[k] = inv
        vk = Mx[k].copy()
        for j in range(5):
            if j != k:
                Mx[j] += mult[j] * vk
        Mx[k] = vk * inv
        Mx[:, k] = colk
    return Mx, perm


def apply_inv(Dinv, perm, t):
    t = t.copy()
    for k in range(5):
        if perm[k] != k:
            t[[k, perm[k]]] = t[[perm[k], k]]
    return Dinv @ t


class BlockChain:
    """One segment: interior blocks 0..L-1 with couplings Lo[k] (to k-1; Lo[0] to the left separator) and Up[k] (to k+1; Up[L-1] to the
    right separator)."""

    def factor(self, Lo, Dg, Up):
        L = Dg.shape[0]
        self.L = L
        self.Lo = Lo
        self.Dinv = np.zeros((L, 5, 5)); self.perm = np.zeros((L, 5), dtype=int); self.G = np.zeros((L, 5, 5))
        YL = np.zeros((L, 5, 5))
        Gp = np.zeros((5, 5)); Yp = np.zeros((5, 5))
        for k in range(L):
            W = np.zeros((5, 15))
            W[:, 0:5] = Dg[k] - (Lo[k] @ Gp if k > 0 else 0.0)
            W[:, 5:10] = Up[k]
            W[:, 10:15] = Lo[0] if k == 0 else -(Lo[k] @ Yp)
            W, perm = gj5(W)
            self.Dinv[k], self.perm[k], self.G[k], YL[k] = W[:, 0:5], perm, W[:, 5:10], W[:, 10:15]
            Gp, Yp = self.G[k], YL[k]
        # backward pass for the spikes
        self.XL = np.zeros((L, 5, 5)); self.XR = np.zeros((L, 5, 5))
        self.XL[L - 1] = YL[L - 1]; self.XR[L - 1] = self.G[L - 1]
        for k in range(L - 2, -1, -1):
            self.XL[k] = YL[k] - self.G[k] @ self.XL[k + 1]
            self.XR[k] = -self.G[k] @ self.XR[k + 1]

    def solve0(self, rhs):
        """X0 = A_int^-1 rhs, rhs [L, 5]."""
        L = self.L
        y = np.zeros((L, 5))
        for k in range(L):
            t = rhs[k] - (self.Lo[k] @ y[k - 1] if k > 0 else 0.0)
            y[k] = apply_inv(self.Dinv[k], self.perm[k], t)
        x = np.zeros((L, 5))
        x[L - 1] = y[L - 1]
        for k in range(L - 2, -1, -1):
            x[k] = y[k] - self.G[k] @ x[k + 1]
        return x


class KktFactor:
    def __init__(self, Lo, Dg, Up, nseg):
        n = Dg.shape[0]
        self.n, self.nseg = n, nseg
        self.sep = [(j * n) // nseg for j in range(nseg)]
        self.Lo, self.Dg, self.Up = Lo, Dg, Up
        self.seg = []
        RD = np.zeros((nseg, 5, 5)); RL = np.zeros((nseg, 5, 5)); RU = np.zeros((nseg, 5, 5))
        for j in range(nseg):
            s0 = self.sep[j]
            s1 = self.sep[j + 1] if j + 1 < nseg else self.sep[0] + n
            idx = [(s0 + 1 + k) % n for k in range(s1 - s0 - 1)]
            ch = BlockChain()
            ch.factor(Lo[idx], Dg[idx], Up[idx])
            self.seg.append((idx, ch))
        for j in range(nseg):
            s0 = self.sep[j]
            idx, ch = self.seg[j]
            idp, chp = self.seg[(j - 1) % nseg]
            # row j: Dg xs_j + Up x_first(j) + Lo x_last(j-1) = r;  x_first(j) = X0 - XL xs_j - XR xs_j+1;  x_last(j-1) = X0 - XL xs_j-1 - XR xs_j
            RD[j] += Dg[s0] - Up[s0] @ ch.XL[0] - Lo[s0] @ chp.XR[-1]
            RU[j] += -Up[s0] @ ch.XR[0]
            RL[j] += -Lo[s0] @ chp.XL[-1]
        self.RD, self.RL, self.RU = RD, RL, RU
        # separators: block cyclic tridiagonal (RL[j] to j-1, RU[j] to j+1).  nseg == 1: everything lands on the diagonal; nseg == 2: both
        # off-diagonal blocks point at the same neighbour.
        if nseg == 1:
            self.R0 = RD[0] + RL[0] + RU[0]
            self.red = None
        elif nseg == 2:
            self.red = None
            self.Rd = np.block([[RD[0], RU[0] + RL[0]], [RL[1] + RU[1], RD[1]]])
        else:
            ch = BlockChain()
            ch.factor(RL[1:], RD[1:], RU[1:])
            self.red = ch
            # separator 0: RD0 x0 + RU0 x_1 + RL0 x_(nseg-1) = r0;  x_1 = X0[0] - (XL[0] + XR[0]) x0 (both spikes are separator 0)
            self.R0 = RD[0] - RU[0] @ (ch.XL[0] + ch.XR[0]) - RL[0] @ (ch.XL[-1] + ch.XR[-1])

    def solve(self, r):
        n, nseg = self.n, self.nseg
        x = np.zeros(n)
        X0s = []
        rr = np.zeros((nseg, 5))
        for j in range(nseg):
            idx, ch = self.seg[j]
            rhs = np.zeros((len(idx), 5)); rhs[:, 0] = r[idx]
            X0s.append(ch.solve0(rhs))
        for j in range(nseg):
            s0 = self.sep[j]
            rr[j, 0] = r[s0]
            rr[j] -= self.Up[s0] @ X0s[j][0] + self.Lo[s0] @ X0s[(j - 1) % nseg][-1]
        if nseg == 1:
            xs = np.linalg.solve(self.R0, rr[0])[None]
        elif nseg == 2:
            xs = np.linalg.solve(self.Rd, rr.reshape(10)).reshape(2, 5)
        else:
            X0 = self.red.solve0(rr[1:])
            r0 = rr[0] - self.RU[0] @ X0[0] - self.RL[0] @ X0[-1]
            x0 = apply_inv(*gj5(self.R0)[:2], r0) if False else np.linalg.solve(self.R0, r0)
            xs = np.zeros((nseg, 5))
            xs[0] = x0
            xs[1:] = X0 - np.einsum("kij,j->ki", self.red.XL + self.red.XR, x0)
        for j in range(nseg):
            idx, ch = self.seg[j]
            x[self.sep[j]] = xs[j, 0]
            x[idx] = X0s[j][:, 0] - ch.XL[:, 0, :] @ xs[j] - ch.XR[:, 0, :] @ xs[(j + 1) % nseg]
        return x


def run2(name, seed=0, nseg=16, sig_lo=-8, sig_hi=12, pin_frac=0.15, kappa_w=False, nsub=None):
    z = np.load(f"/root/repo/tests/golden/{name}.npz")
    ref, nv, s = z["reftrack"], z["normvec"], z["scaling"]
    n = ref.shape[0]
    rng = np.random.default_rng(seed)
    T, R, a, b, Eb, k_ref = kkt_coeffs(ref[:, :2], nv, s)
    Td, Rd = tri_dense(T, n), tri_dense(R, n)
    Es = np.diag(a) @ np.linalg.solve(Td, Rd @ np.diag(nv[:, 0])) + np.diag(b) @ np.linalg.solve(Td, Rd @ np.diag(nv[:, 1]))
    sig = 10.0 ** rng.uniform(sig_lo, sig_hi, n)
    pinned = rng.uniform(size=n) < pin_frac
    w = 1.0 + (10.0 ** rng.uniform(-3, 10, n) if kappa_w else 0.0) * np.ones(n)
    r = rng.standard_normal(n)
    M = Es.T @ (w[:, None] * Es) + np.diag(sig)
    M[pinned, :] = 0.0; M[:, pinned] = 0.0; M[pinned, pinned] = 1.0
    x_ch = banded_chol_solve(M, r)
    Lo, Dg, Up = kkt_blocks(T, R, a, b, nv, sig, w, pinned)
    F = KktFactor(Lo, Dg, Up, nseg)
    x_k = F.solve(r)
    scale = np.abs(x_ch).max()
    res = lambda x: np.abs(M @ x - r).max() / (np.abs(M) @ np.abs(x) + np.abs(r)).max()
    print(f"{name} n {n} nseg {nseg} seed {seed}: cond(M) {np.linalg.cond(M):.1e}  backward error cholesky {res(x_ch):.2e} kkt {res(x_k):.2e}   kkt vs cholesky {np.abs(x_k - x_ch).max() / scale:.2e}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "kernel":
    for nm in ("rounded_rectangle", "handling_track", "berlin_2018_n333"):
        for nseg in (1, 2, 3, 7, 16):
            run2(nm, nseg, nseg=nseg)
        run2(nm, 2, kappa_w=True)
        run2(nm, 3, sig_lo=-12, sig_hi=-6, pin_frac=0.0)
        run2(nm, 4, pin_frac=0.9)
    for nm in ("oval_n2000", "oval_n2000_c5"):
        run2(nm, 0); run2(nm, 2, kappa_w=True); run2(nm, 3, sig_lo=-12, sig_hi=-6, pin_frac=0.0); run2(nm, 5, sig_lo=-3, sig_hi=3, pin_frac=0.3)


# ----------------------------------------------------------------------------------------------------------------------
# Round 3, late: the separators' block-cyclic system by PARALLEL CYCLIC REDUCTION (nseg a power of two) instead of one chain of nseg - 1
# steps -- log2(nseg) levels, every separator eliminated against both neighbours at once:
#     L_j x_(j-s) + D_j x_j + U_j x_(j+s) = b_j,      a_j = L_j D_(j-s)^-1,  c_j = U_j D_(j+s)^-1
#     D_j <- D_j - a_j U_(j-s) - c_j L_(j+s),   L_j <- -a_j L_(j-s),   U_j <- -c_j U_(j+s),   b_j <- b_j - a_j b_(j-s) - c_j b_(j+s)
# and after the last level x_j = (D_j + L_j + U_j)^-1 b_j.  The 5 x 5 inverses use the kernel's STATIC pivot order.
#   python scripts/proto_kkt.py pcr
# ----------------------------------------------------------------------------------------------------------------------
PIV_ROWS, PIV_COLS = (3, 4, 1, 2, 0), (1, 2, 3, 4, 0)


def inv5_static(A):
    """Gauss-Jordan inverse with the pivot positions the kernel uses (no search)."""
    W = np.hstack([A.astype(float).copy(), np.eye(5)])
    for r, c in zip(PIV_ROWS, PIV_COLS):
        W[r] = W[r] / W[r, c]
        for i in range(5):
            if i != r:
                W[i] = W[i] - W[i, c] * W[r]
    # row r now holds the unit vector e_c: x_c = (row r of the right half)
    out = np.zeros((5, 5))
    for r, c in zip(PIV_ROWS, PIV_COLS):
        out[c] = W[r, 5:]
    return out


class PcrSeparators:
    def __init__(self, RD, RL, RU):
        N = RD.shape[0]
        assert N & (N - 1) == 0
        D, L, U = RD.copy(), RL.copy(), RU.copy()
        self.N = N
        self.lev = []
        s = 1
        while s < N:
            Di = np.array([inv5_static(D[j]) for j in range(N)])
            a = np.array([L[j] @ Di[(j - s) % N] for j in range(N)])
            c = np.array([U[j] @ Di[(j + s) % N] for j in range(N)])
            Dn = np.array([D[j] - a[j] @ U[(j - s) % N] - c[j] @ L[(j + s) % N] for j in range(N)])
            Ln = np.array([-a[j] @ L[(j - s) % N] for j in range(N)])
            Un = np.array([-c[j] @ U[(j + s) % N] for j in range(N)])
            self.lev.append((s, a, c))
            D, L, U = Dn, Ln, Un
            s *= 2
        self.F = np.array([inv5_static(D[j] + L[j] + U[j]) for j in range(N)])

    def solve(self, b):
        N = self.N
        b = b.copy()
        for s, a, c in self.lev:
            b = np.array([b[j] - a[j] @ b[(j - s) % N] - c[j] @ b[(j + s) % N] for j in range(N)])
        return np.einsum("jik,jk->ji", self.F, b)


def run_pcr(name, seed=0, nseg=16, sig_lo=-8, sig_hi=12, pin_frac=0.15, kappa_w=False):
    z = np.load(f"/root/repo/tests/golden/{name}.npz")
    ref, nv, s = z["reftrack"], z["normvec"], z["scaling"]
    n = ref.shape[0]
    rng = np.random.default_rng(seed)
    T, R, a, b, Eb, k_ref = kkt_coeffs(ref[:, :2], nv, s)
    Td, Rd = tri_dense(T, n), tri_dense(R, n)
    Es = np.diag(a) @ np.linalg.solve(Td, Rd @ np.diag(nv[:, 0])) + np.diag(b) @ np.linalg.solve(Td, Rd @ np.diag(nv[:, 1]))
    sig = 10.0 ** rng.uniform(sig_lo, sig_hi, n)
    pinned = rng.uniform(size=n) < pin_frac
    w = 1.0 + (10.0 ** rng.uniform(-3, 10, n) if kappa_w else 0.0) * np.ones(n)
    r = rng.standard_normal(n)
    M = Es.T @ (w[:, None] * Es) + np.diag(sig)
    M[pinned, :] = 0.0; M[:, pinned] = 0.0; M[pinned, pinned] = 1.0
    Lo, Dg, Up = kkt_blocks(T, R, a, b, nv, sig, w, pinned)
    F = KktFactor(Lo, Dg, Up, nseg)
    x_chain = F.solve(r)
    # the same segments, the separators by PCR
    P = PcrSeparators(F.RD, F.RL, F.RU)
    X0s = []
    rr = np.zeros((nseg, 5))
    for j in range(nseg):
        idx, ch = F.seg[j]
        rhs = np.zeros((len(idx), 5)); rhs[:, 0] = r[idx]
        X0s.append(ch.solve0(rhs))
    for j in range(nseg):
        s0 = F.sep[j]
        rr[j, 0] = r[s0]
        rr[j] -= Up[s0] @ X0s[j][0] + Lo[s0] @ X0s[(j - 1) % nseg][-1]
    xs = P.solve(rr)
    x = np.zeros(n)
    for j in range(nseg):
        idx, ch = F.seg[j]
        x[F.sep[j]] = xs[j, 0]
        x[idx] = X0s[j][:, 0] - ch.XL[:, 0, :] @ xs[j] - ch.XR[:, 0, :] @ xs[(j + 1) % nseg]
    res = lambda x: np.abs(M @ x - r).max() / (np.abs(M) @ np.abs(x) + np.abs(r)).max()
    print(f"{name} n {n} nseg {nseg} seed {seed}: cond(M) {np.linalg.cond(M):.1e}  backward error chain {res(x_chain):.2e}  pcr {res(x):.2e}   pcr vs chain {np.abs(x - x_chain).max() / np.abs(x_chain).max():.2e}")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pcr":
    for nm in ("rounded_rectangle", "handling_track", "berlin_2018_n333"):
        for nseg in (2, 4, 8, 16):
            run_pcr(nm, nseg, nseg=nseg)
        run_pcr(nm, 2, kappa_w=True)
        run_pcr(nm, 3, sig_lo=-12, sig_hi=-6, pin_frac=0.0)
        run_pcr(nm, 4, pin_frac=0.9)
    for nm in ("oval_n2000", "oval_n2000_c5"):
        run_pcr(nm, 0); run_pcr(nm, 2, kappa_w=True); run_pcr(nm, 3, sig_lo=-12, sig_hi=-6, pin_frac=0.0); run_pcr(nm, 5, sig_lo=-3, sig_hi=3, pin_frac=0.3)
