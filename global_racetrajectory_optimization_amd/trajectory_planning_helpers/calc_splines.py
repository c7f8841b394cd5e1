"""
Host-side shim of tph.calc_splines.calc_splines -- boundary [REF helper_funcs_glob/src/prep_track.py:48-51],
[REF main_globaltraj.py:568-569].  Closed tracks only (the only case the reference's mincurv flow exercises).

Instead of the dense 4N x 4N solve of the upstream formulation, the closed cubic-spline conditions are reduced to the
cyclic tridiagonal system in the quadratic coefficients c_i (derivation in DESIGN.md section 3):

    c_i + (2 s_i^2 + 2 s_i) c_{i+1} + s_i s_{i+1}^2 c_{i+2} = 3 (s_i D_{i+1} - D_i),     D_i = p_{i+1} - p_i
    d_i = (s_i^2 c_{i+1} - c_i) / 3,   b_i = D_i - (2 c_i + s_i^2 c_{i+1}) / 3,   a_i = p_i

The dense matrix M is still returned because the reference hands it on to opt_min_curv
[REF main_globaltraj.py:267]; our opt_min_curv only reads the N scalings s_i back out of it.
"""
import numpy as np
from scipy.linalg import solve_banded


def _solve_cyclic_tridiag(sub, diag, sup, rhs):
    """Rows m: sub[m] x[m-1] + diag[m] x[m] + sup[m] x[m+1] = rhs[m] (indices cyclic); rhs [N, k]."""
    n = diag.size
    if n < 3:
        raise ValueError("need at least 3 spline segments")
    ab = np.zeros((3, n))
    ab[0, 1:] = sup[:-1]
    ab[1, :] = diag
    ab[2, :-1] = sub[1:]
    # Woodbury for the two corner entries T[0, n-1] = sub[0], T[n-1, 0] = sup[n-1]
    U = np.zeros((n, 2))
    U[0, 0] = 1.0
    U[n - 1, 1] = 1.0
    V = np.zeros((2, n))
    V[0, n - 1] = sub[0]
    V[1, 0] = sup[n - 1]
    y = solve_banded((1, 1), ab, np.column_stack((rhs, U)))
    yr, yu = y[:, :rhs.shape[1]], y[:, rhs.shape[1]:]
    corr = np.linalg.solve(np.eye(2) + V @ yu, V @ yr)
    return yr - yu @ corr


def spline_scalings(path_cl, el_lengths=None, use_dist_scaling=True):
    n = path_cl.shape[0] - 1
    if not use_dist_scaling:
        return np.ones(n)
    if el_lengths is None:
        el = np.sqrt(np.sum(np.diff(path_cl, axis=0) ** 2, axis=1))
    else:
        el = np.array(el_lengths, dtype=np.float64)
    return el / np.roll(el, -1)


def closed_spline_coeffs(points, scaling):
    """points [N, k] (unclosed), scaling [N] -> a, b, c, d each [N, k]."""
    n = points.shape[0]
    s = scaling
    s_next = np.roll(s, -1)
    delta = np.roll(points, -1, axis=0) - points
    rhs = 3.0 * (s[:, None] * np.roll(delta, -1, axis=0) - delta)
    # equation i couples c_i, c_{i+1}, c_{i+2}; index rows by the centre unknown m = i + 1
    sub = np.roll(np.ones(n), 1)
    diag = np.roll(2.0 * s ** 2 + 2.0 * s, 1)
    sup = np.roll(s * s_next ** 2, 1)
    c = _solve_cyclic_tridiag(sub, diag, sup, np.roll(rhs, 1, axis=0))
    c_next = np.roll(c, -1, axis=0)
    d = (s[:, None] ** 2 * c_next - c) / 3.0
    b = delta - (2.0 * c + s[:, None] ** 2 * c_next) / 3.0
    return points.copy(), b, c, d


def build_les_matrix(n, scaling):
    """The dense 4N x 4N matrix of the upstream formulation -- returned for interface parity."""
    M = np.zeros((4 * n, 4 * n))
    i = np.arange(n)
    j = 4 * i
    M[j, j] = 1.0
    for k in range(4):
        M[j + 1, j + k] = 1.0
    inner = i[:-1]
    ji = 4 * inner
    M[ji + 2, ji + 1], M[ji + 2, ji + 2], M[ji + 2, ji + 3] = 1.0, 2.0, 3.0
    M[ji + 2, ji + 5] = -scaling[:-1]
    M[ji + 3, ji + 2], M[ji + 3, ji + 3] = 2.0, 6.0
    M[ji + 3, ji + 6] = -2.0 * scaling[:-1] ** 2
    M[-2, 1] = scaling[-1]
    M[-2, -3:] = (-1.0, -2.0, -3.0)
    M[-1, 2] = 2.0 * scaling[-1] ** 2
    M[-1, -2:] = (-2.0, -6.0)
    return M


def scalings_from_les_matrix(A, check=True):
    """Recover s_i from the matrix calc_splines returned: the inverse of build_les_matrix.

    The engine never sees `A`: everything opt_min_curv needs from it is the N scalings.  That is only valid if `A` IS the
    matrix of the closed-spline system (what the reference passes on from calc_splines [REF main_globaltraj.py:267,
    prep_track.py:48-51]); `check` verifies it in O(N) + one counting pass -- the structural entries of every 4-row block, the four entries
    that must mirror the scalings (-s, -2 s^2), positivity, and the number of non-zeros (12 N: nothing outside the
    pattern) -- and raises RuntimeError otherwise: a silently wrong alpha is the alternative (there is no dense CPU path in
    the product to fall back to).
    """
    A = np.asarray(A)
    n = A.shape[0] // 4
    i = np.arange(n - 1)
    s = np.empty(n)
    s[:-1] = -A[4 * i + 2, 4 * i + 5]
    s[-1] = A[4 * n - 2, 1]
    if check:
        j = 4 * np.arange(n)
        ji = 4 * i
        ok = bool(np.all(np.isfinite(s)) and np.all(s > 0.0))
        ok = ok and bool(np.all(A[j, j] == 1.0) and np.all(A[j + 1, j] == 1.0) and np.all(A[j + 1, j + 1] == 1.0)
                         and np.all(A[j + 1, j + 2] == 1.0) and np.all(A[j + 1, j + 3] == 1.0))
        ok = ok and bool(np.all(A[ji + 2, ji + 1] == 1.0) and np.all(A[ji + 2, ji + 2] == 2.0) and np.all(A[ji + 2, ji + 3] == 3.0)
                         and np.all(A[ji + 3, ji + 2] == 2.0) and np.all(A[ji + 3, ji + 3] == 6.0))
        ok = ok and bool(np.all(A[-2, -3:] == (-1.0, -2.0, -3.0)) and np.all(A[-1, -2:] == (-2.0, -6.0)))
        if ok:
            s2 = 2.0 * s * s
            ok = bool(np.allclose(-A[ji + 3, ji + 6], s2[:-1], rtol=1e-12, atol=0.0) and np.isclose(A[-1, 2], s2[-1], rtol=1e-12, atol=0.0))
        if ok:
            # the 12 N entries checked above are all there is: one pass over A, no temporaries (~60 ms at N = 2000, where
            # upstream spends 13 s inverting the same matrix)
            ok = int(np.count_nonzero(A)) == 12 * n
        if not ok:
            raise RuntimeError("Spline equation system matrix A does not have the structure of calc_splines' closed-spline "
                               "system (the MI355X engine derives everything from the N spline scalings it encodes and "
                               "cannot use an arbitrary matrix)")
    return s


def calc_splines(path: np.ndarray, el_lengths: np.ndarray = None, psi_s: float = None, psi_e: float = None,
                 use_dist_scaling: bool = True) -> tuple:
    path = np.asarray(path, dtype=np.float64)
    closed = bool(np.all(np.isclose(path[0], path[-1]))) and psi_s is None
    if not closed:
        raise NotImplementedError("calc_splines shim: only closed paths (first point repeated at the end) are supported")
    if el_lengths is not None and path.shape[0] != el_lengths.size + 1:
        raise RuntimeError("el_lengths input must be one element smaller than path input!")
    n = path.shape[0] - 1
    scaling = spline_scalings(path, el_lengths, use_dist_scaling)
    a, b, c, d = closed_spline_coeffs(path[:-1], scaling)
    coeffs_x = np.column_stack((a[:, 0], b[:, 0], c[:, 0], d[:, 0]))
    coeffs_y = np.column_stack((a[:, 1], b[:, 1], c[:, 1], d[:, 1]))
    normvec = np.stack((coeffs_y[:, 1], -coeffs_x[:, 1]), axis=1)
    normvec_normalized = normvec / np.sqrt(np.sum(normvec ** 2, axis=1))[:, None]
    return coeffs_x, coeffs_y, build_les_matrix(n, scaling), normvec_normalized
