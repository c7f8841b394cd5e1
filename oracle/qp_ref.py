"""
ORACLE -- TEST INFRASTRUCTURE ONLY (header of oracle/tph_ref.py applies).  PARITY UNPINNED by the reference.

QP routes used to pin the oracle against itself:
  solve_qp_gi      dense Goldfarb-Idnani (oracle/gi_dense.c), quadprog calling convention
                   quadprog.solve_qp(H, -f, -G.T, -h, 0)[0]  
  solve_qp_gi_zero_width_as_equalities   the same solver with exactly dependent box pairs (lo = hi) stated as equalities (meq), see there
  solve_box_bvls   scipy.optimize.lsq_linear(E, -2 k_ref, bounds) -- valid when the kappa rows are inactive
  solve_box_second_route   the same least-squares form by the trust-region-reflective method (an interior method on the dense
                   E with exact SVD steps: neither an active-set method nor the normal equations) -- the independent route at
                   the sizes where BVLS is too slow (N = 663 / 776 / 2000)
  kkt_residuals    certificate: stationarity / primal / dual feasibility / complementarity
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE, "libgi_dense.so"], check=True)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgi_dense.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        _LIB.gi_dense_solve.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, ctypes.c_int, dp, dp, ip, ip, ip, dp]
        _LIB.gi_dense_solve.restype = ctypes.c_int
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def solve_qp_quadprog_convention(Gm, a, C, b, meq=0):
    """min 1/2 x'Gm x - a'x  s.t.  C'x >= b.  Returns (x, f, lagr, iact, iters) like quadprog (minus xu)."""
    n = Gm.shape[0]
    m = C.shape[1]
    Gm = np.ascontiguousarray(Gm, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    Ccm = np.ascontiguousarray(C.T, dtype=np.float64)      # column-major n x m == row-major m x n
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    lagr = np.zeros(m)
    iact = np.zeros(n + 1, dtype=np.int32)
    nact = ctypes.c_int(0)
    iters = np.zeros(2, dtype=np.int32)
    fval = ctypes.c_double(0.0)
    st = _lib().gi_dense_solve(n, m, _dp(Gm), _dp(a), _dp(Ccm), _dp(b), meq, _dp(x), _dp(lagr),
                               iact.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(nact),
                               iters.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(fval))
    if st == 2:
        raise ValueError("matrix G is not positive definite")
    if st == 1:
        raise ValueError("constraints are inconsistent, no solution")
    if st != 0:
        raise RuntimeError("gi_dense_solve failed with status %d" % st)
    return x, fval.value, lagr, iact[:nact.value].copy(), iters


def solve_qp_gi(H, f, G, h, info=None):
    """min 1/2 a'Ha + f'a  s.t.  G a <= h  through the quadprog convention (H, -f, -G.T, -h)."""
    x, fv, lagr, iact, iters = solve_qp_quadprog_convention(H, -f, -G.T, -h, 0)
    if info is not None:
        info.update(f=fv, lagr=lagr, iact=iact, iters=iters)
    return x


def solve_qp_gi_zero_width_as_equalities(H, f, G, h, info=None):
    """The same QP with every EXACTLY DEPENDENT box pair (rows i and n + i of upstream's G = [I; -I; E; -E] with h_i = -h_(n+i): a waypoint
    whose corridor is exactly as wide as the vehicle, lo = hi) stated the way quadprog wants an equality stated -- one row, counted in `meq` --
    instead of the two opposite inequalities tph builds.  Still oracle/gi_dense.c, i.e. qpgen2's algorithm.

    Why it exists (round 6, found by tests/test_gpu_parity.py::test_pinned_variables_and_bad_input after gi_dense.c became qpgen2's rule set): on
    the two-inequality form quadprog's outcome is decided by a rounding residue.  After one row of the pair has entered, later steps move x along
    directions orthogonal to its normal only up to rounding; when the partner's slack drifts below -vsmall (1.4e-15; seen: -4.5e-15) the
    partner is "violated", its normal is minus an active one, so z = 0, r = -1 for the active partner, a chain of dual steps drops every
    constraint with r > 0 and the solve ends in "constraints are inconsistent, no solution" -- on a feasible QP.  When the residue stays below
    vsmall the slack is set to zero and the solve ends at the optimum (tests/test_emu_gi.py's case).  The engine returns the optimum in both
    cases: a stated deviation from what quadprog would do on such input, pinned by this form and by the BVLS route."""
    n = H.shape[0]
    pin = np.where(h[:n] + h[n:2 * n] == 0.0)[0]
    keep = np.ones(G.shape[0], dtype=bool)
    keep[pin] = False
    keep[n + pin] = False
    C = np.hstack((-G[pin].T, -G[keep].T))
    b = np.concatenate((-h[pin], -h[keep]))
    x, fv, lagr, iact, iters = solve_qp_quadprog_convention(H, -f, C, b, meq=int(pin.size))
    if info is not None:
        info.update(f=fv, lagr=lagr, iact=iact, iters=iters, pinned=pin)
    return x


def solve_box_bvls(E, k_ref, lo, hi, f_scale=2.0):
    """Box-only route: 1/2 a'E'Ea + (f_scale E'k_ref)'a == 1/2 ||E a + f_scale k_ref||^2 + const."""
    from scipy.optimize import lsq_linear
    res = lsq_linear(E, -f_scale * k_ref, bounds=(lo, hi), method="bvls", tol=1e-15, max_iter=50 * E.shape[1])
    return res.x


def solve_box_second_route(E, k_ref, lo, hi, f_scale=2.0):
    """Second, independent route for the box-only QP at any size: scipy's trust-region-reflective bounded least squares
    on  1/2 ||E a + f_scale k_ref||^2  (dense E, exact least-squares steps).  Shares nothing with the Goldfarb-Idnani
    oracle: no Cholesky of H = E'E (cond 1e9..1e12; this route sees cond(E) = its square root), no active-set logic.
    Valid whenever the curvature rows are inactive at its optimum (true on every fixture; the caller checks)."""
    from scipy.optimize import lsq_linear
    res = lsq_linear(E, -f_scale * k_ref, bounds=(lo, hi), method="trf", tol=1e-15, lsq_solver="exact", max_iter=1000)
    return res.x


def kkt_residuals(H, f, G, h, x, act_tol=1e-9):
    """Certificate for min 1/2 x'Hx + f'x s.t. Gx <= h: multipliers by NNLS on the near-active rows."""
    from scipy.optimize import nnls
    g = H @ x + f
    s = h - G @ x
    act = np.where(s <= act_tol)[0]
    if act.size:
        lam, _ = nnls(G[act].T, -g, maxiter=20 * act.size + 100)
        stat = g + G[act].T @ lam
    else:
        lam = np.zeros(0)
        stat = g
    scale = max(1.0, float(np.max(np.abs(H))) * max(1.0, float(np.max(np.abs(x)))))
    return dict(stationarity=float(np.max(np.abs(stat))) / scale, stationarity_abs=float(np.max(np.abs(stat))),
                primal=float(max(0.0, -np.min(s))), n_active=int(act.size), lam=lam, act=act)
