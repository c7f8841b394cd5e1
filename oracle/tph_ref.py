"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not shipped, not on the product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (global_racetrajectory_optimization_amd/) never does.

PARITY UNPINNED BY THE REFERENCE: the arithmetic of the hot path lives in the third-party
packages trajectory_planning_helpers==0.76 [REF requirements.txt:3] and quadprog
[REF Readme.md:40,44]; neither is vendored under /root/reference nor installable here, and the
reference ships no tests / golden vectors.  This file is a
dense-faithful CPU restatement of the published algorithm (Heilmeier et al., "Minimum Curvature
Trajectory Planning and Control for an Autonomous Racecar", DOI 10.1080/00423114.2019.1631455,
cited at [REF Readme.md:133-137]) following the matrix conventions recorded in SURVEY.md App. A,
anchored on the reference's own call sites:

  opt_min_curv   <- [REF main_globaltraj.py:264-271, 344-350]
  iqp_handler    <- [REF main_globaltraj.py:273-284]
  calc_splines   <- [REF helper_funcs_glob/src/prep_track.py:48-51]
  create_raceline<- [REF main_globaltraj.py:371-376]
  opt_shortest_path <- [REF main_globaltraj.py:286-290]

"Dense-faithful" means: no structure is exploited.  The 4N x 4N spline system is built and
inverted densely, H/f/E_kappa are formed with dense products, and the QP is handed with all 4N
inequality rows to a dense Goldfarb-Idnani solver (oracle/gi_dense.c, a stand-in for quadprog).
We pin the oracle ourselves by two independent solution routes (dense GI vs. scipy BVLS on the
least-squares form) plus a KKT certificate -- see tests/test_oracle.py and tests/golden/.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------------------------------------------------
# closed cubic spline through points: dense 4N x 4N linear equation system
# ----------------------------------------------------------------------------------------------------------------------

def calc_splines(path, el_lengths=None, use_dist_scaling=True):
    """Closed cubic splines through `path` (first point repeated at the end).

    Boundary [REF helper_funcs_glob/src/prep_track.py:48-51] (called with `path=` only ->
    distance scaling on) and inside iqp_handler (use_dist_scaling=False).
    Unknowns z = [a_0 b_0 c_0 d_0 a_1 ...]; returns (coeffs_x[N,4], coeffs_y[N,4], M[4N,4N],
    normvec_normalized[N,2]); normals = tangent rotated clockwise (pointing right,
    [REF Readme.md:80-81]).
    """
    path = np.asarray(path, dtype=np.float64)
    if not np.all(np.isclose(path[0], path[-1])):
        raise NotImplementedError("oracle restates the closed-track case only (the only one the reference uses)")
    n = path.shape[0] - 1

    if use_dist_scaling:
        if el_lengths is None:
            el = np.sqrt(np.sum(np.diff(path, axis=0) ** 2, axis=1))
        else:
            el = np.array(el_lengths, dtype=np.float64)
        el = np.append(el, el[0])
        scaling = el[:-1] / el[1:]
    else:
        scaling = np.ones(n)

    M = np.zeros((4 * n, 4 * n))
    bx = np.zeros(4 * n)
    by = np.zeros(4 * n)
    for i in range(n):
        j = 4 * i
        M[j, j] = 1.0                                   # a_i = p_i
        M[j + 1, j:j + 4] = 1.0                         # a_i + b_i + c_i + d_i = p_{i+1}
        bx[j], bx[j + 1] = path[i, 0], path[i + 1, 0]
        by[j], by[j + 1] = path[i, 1], path[i + 1, 1]
        if i < n - 1:
            M[j + 2, j + 1:j + 4] = (1.0, 2.0, 3.0)     # heading continuity
            M[j + 2, j + 5] = -scaling[i]
            M[j + 3, j + 2:j + 4] = (2.0, 6.0)          # curvature continuity
            M[j + 3, j + 6] = -2.0 * scaling[i] ** 2
        else:                                           # wrap-around rows carry the opposite sign upstream
            M[j + 2, 1] = scaling[-1]
            M[j + 2, j + 1:j + 4] = (-1.0, -2.0, -3.0)
            M[j + 3, 2] = 2.0 * scaling[-1] ** 2
            M[j + 3, j + 2:j + 4] = (-2.0, -6.0)

    coeffs_x = np.linalg.solve(M, bx).reshape(n, 4)
    coeffs_y = np.linalg.solve(M, by).reshape(n, 4)
    normvec = np.stack((coeffs_y[:, 1], -coeffs_x[:, 1]), axis=1)
    normvec /= np.sqrt(np.sum(normvec ** 2, axis=1))[:, None]
    return coeffs_x, coeffs_y, M, normvec


# ----------------------------------------------------------------------------------------------------------------------
# opt_min_curv assembly, dense
# ----------------------------------------------------------------------------------------------------------------------

F_SCALE = 2.0   # the factor-2 quirk of SURVEY.md App. A.4: quadprog sees 1/2 a'Ha + f'a with f = 2 E' k_ref


def assemble_dense(reftrack, normvectors, A):
    """H, f, E_kappa, k_ref and the pieces the post-check needs -- dense, no structure used."""
    reftrack = np.asarray(reftrack, dtype=np.float64)
    normvectors = np.asarray(normvectors, dtype=np.float64)
    n = reftrack.shape[0]
    if n != normvectors.shape[0]:
        raise RuntimeError("Array size of reftrack should be the same as normvectors!")
    if A.shape[0] != 4 * n or A.shape[0] != A.shape[1]:
        raise RuntimeError("Spline equation system matrix A has wrong dimensions!")

    A_ex_b = np.zeros((n, 4 * n))
    A_ex_c = np.zeros((n, 4 * n))
    A_ex_b[np.arange(n), 4 * np.arange(n) + 1] = 1.0
    A_ex_c[np.arange(n), 4 * np.arange(n) + 2] = 2.0

    A_inv = np.linalg.inv(A)
    T_c = A_ex_c @ A_inv
    T_b = A_ex_b @ A_inv

    M_x = np.zeros((4 * n, n))
    M_y = np.zeros((4 * n, n))
    q_x = np.zeros(4 * n)
    q_y = np.zeros(4 * n)
    for i in range(n):
        nxt = (i + 1) % n
        M_x[4 * i, i], M_x[4 * i + 1, nxt] = normvectors[i, 0], normvectors[nxt, 0]
        M_y[4 * i, i], M_y[4 * i + 1, nxt] = normvectors[i, 1], normvectors[nxt, 1]
        q_x[4 * i], q_x[4 * i + 1] = reftrack[i, 0], reftrack[nxt, 0]
        q_y[4 * i], q_y[4 * i + 1] = reftrack[i, 1], reftrack[nxt, 1]

    x_p = T_b @ q_x
    y_p = T_b @ q_y
    den = (x_p ** 2 + y_p ** 2) ** 1.5
    c = np.divide(1.0, den, out=np.zeros_like(den), where=den != 0)
    P_xx = np.diag(c ** 2 * y_p ** 2)
    P_yy = np.diag(c ** 2 * x_p ** 2)
    P_xy = np.diag(-2.0 * c ** 2 * x_p * y_p)

    T_nx = T_c @ M_x
    T_ny = T_c @ M_y
    H = T_nx.T @ P_xx @ T_nx + T_ny.T @ P_xy @ T_nx + T_ny.T @ P_yy @ T_ny
    H = 0.5 * (H + H.T)

    tcqx = T_c @ q_x
    tcqy = T_c @ q_y
    f = (F_SCALE * tcqx @ P_xx @ T_nx + tcqx @ P_xy @ T_ny + tcqy @ P_xy @ T_nx + F_SCALE * tcqy @ P_yy @ T_ny)

    Q_x = np.diag(c * y_p)
    Q_y = np.diag(c * x_p)
    E_kappa = Q_y @ T_ny - Q_x @ T_nx
    k_ref = Q_y @ tcqy - Q_x @ tcqx
    aux = dict(T_b=T_b, T_c=T_c, M_x=M_x, M_y=M_y, q_x=q_x, q_y=q_y, x_p=x_p, y_p=y_p, T_nx=T_nx, T_ny=T_ny)
    return H, f, E_kappa, k_ref, aux


def constraints_dense(reftrack, E_kappa, k_ref, kappa_bound, w_veh):
    """G (4N x N), h (4N) in the upstream row order [I; -I; E; -E]."""
    n = reftrack.shape[0]
    dev_max_right = reftrack[:, 2] - w_veh / 2
    dev_max_left = reftrack[:, 3] - w_veh / 2
    if np.any(-dev_max_right > dev_max_left) or np.any(-dev_max_left > dev_max_right):
        raise RuntimeError("Problem not solvable, track might be too small to run with current safety distance!")
    G = np.vstack((np.eye(n), -np.eye(n), E_kappa, -E_kappa))
    h = np.concatenate((dev_max_right, dev_max_left, kappa_bound - k_ref, kappa_bound + k_ref))
    return G, h


def curv_error(alpha, aux):
    """Post-check of SURVEY.md App. A.5: max |kappa(lin. at solution) - kappa(lin. at reference)|."""
    q_x_t = aux["q_x"] + aux["M_x"] @ alpha
    q_y_t = aux["q_y"] + aux["M_y"] @ alpha
    x_p_t = aux["T_b"] @ q_x_t
    y_p_t = aux["T_b"] @ q_y_t
    x_pp = aux["T_c"] @ aux["q_x"] + aux["T_nx"] @ alpha
    y_pp = aux["T_c"] @ aux["q_y"] + aux["T_ny"] @ alpha
    x_p, y_p = aux["x_p"], aux["y_p"]
    k_orig = (x_p * y_pp - y_p * x_pp) / (x_p ** 2 + y_p ** 2) ** 1.5
    k_sol = (x_p_t * y_pp - y_p_t * x_pp) / (x_p_t ** 2 + y_p_t ** 2) ** 1.5
    return float(np.max(np.abs(k_sol - k_orig)))


def opt_min_curv(reftrack, normvectors, A, kappa_bound, w_veh, solver=None, return_internals=False):
    """Dense-faithful restatement of tph.opt_min_curv.opt_min_curv (closed tracks).

    Boundary [REF main_globaltraj.py:264-271]; returns (alpha[N], curv_error_max).
    `solver(H, f, G, h) -> alpha` minimises 1/2 a'Ha + f'a s.t. G a <= h; default = dense
    Goldfarb-Idnani in oracle/gi_dense.c (quadprog stand-in).
    """
    H, f, E_kappa, k_ref, aux = assemble_dense(reftrack, normvectors, A)
    G, h = constraints_dense(np.asarray(reftrack, dtype=np.float64), E_kappa, k_ref, kappa_bound, w_veh)
    if solver is None:
        from oracle import qp_ref
        solver = qp_ref.solve_qp_gi
    alpha = solver(H, f, G, h)
    err = curv_error(alpha, aux)
    if return_internals:
        return alpha, err, dict(H=H, f=f, E=E_kappa, k_ref=k_ref, G=G, h=h)
    return alpha, err


# ----------------------------------------------------------------------------------------------------------------------
# shortest path
# ----------------------------------------------------------------------------------------------------------------------

def shortest_path_dense(reftrack, normvectors, w_veh):
    """Dense H, f, G, h of tph.opt_shortest_path (trajectory_planning_helpers==0.76, not vendored; restated from its
    published form, call site [REF main_globaltraj.py:286-290]).  Point by point as upstream accumulates it: every
    segment i -> i+1 (the last one closing the ring) adds the square of its shifted length
    |p_{i+1} + a_{i+1} n_{i+1} - p_i - a_i n_i|^2 to the cost 1/2 a'Ha + f'a (+ const)."""
    reftrack = np.asarray(reftrack, dtype=np.float64)
    nv = np.asarray(normvectors, dtype=np.float64)
    n = reftrack.shape[0]
    H = np.zeros((n, n))
    f = np.zeros(n)
    for i in range(n):
        j = i + 1 if i < n - 1 else 0
        H[i, i] += 2.0 * (nv[i, 0] ** 2 + nv[i, 1] ** 2)
        H[j, j] += 2.0 * (nv[j, 0] ** 2 + nv[j, 1] ** 2)
        hij = -2.0 * (nv[i, 0] * nv[j, 0] + nv[i, 1] * nv[j, 1])
        H[i, j] += hij
        H[j, i] += hij
        f[i] += 2.0 * (nv[i, 0] * (reftrack[i, 0] - reftrack[j, 0]) + nv[i, 1] * (reftrack[i, 1] - reftrack[j, 1]))
        f[j] += 2.0 * (nv[j, 0] * (reftrack[j, 0] - reftrack[i, 0]) + nv[j, 1] * (reftrack[j, 1] - reftrack[i, 1]))
    dev_max_right = reftrack[:, 2] - w_veh / 2.0
    dev_max_left = reftrack[:, 3] - w_veh / 2.0
    dev_max_right[dev_max_right < 0.001] = 0.001       # upstream clips instead of rejecting
    dev_max_left[dev_max_left < 0.001] = 0.001
    G = np.vstack((np.eye(n), -np.eye(n)))
    h = np.append(dev_max_right, dev_max_left)
    return H, f, G, h


def opt_shortest_path(reftrack, normvectors, w_veh, solver=None, return_internals=False):
    """Restatement of tph.opt_shortest_path.opt_shortest_path: alpha [N] from the dense QP, solved by the dense
    Goldfarb-Idnani stand-in for quadprog (oracle/gi_dense.c)."""
    H, f, G, h = shortest_path_dense(reftrack, normvectors, w_veh)
    if solver is None:
        from oracle import qp_ref
        solver = qp_ref.solve_qp_gi
    alpha = solver(H, f, G, h)
    if return_internals:
        return alpha, dict(H=H, f=f, G=G, h=h)
    return alpha


def path_length_sq(reftrack, normvectors, alpha):
    """Sum of squared segment lengths of the closed polygon through p_i + alpha_i n_i (what shortest_path_dense encodes)."""
    p = np.asarray(reftrack, dtype=np.float64)[:, :2] + np.asarray(alpha)[:, None] * np.asarray(normvectors, dtype=np.float64)
    dp = np.roll(p, -1, axis=0) - p
    return float(np.sum(dp * dp))


# ----------------------------------------------------------------------------------------------------------------------
# glue inside the IQP loop
# ----------------------------------------------------------------------------------------------------------------------

def calc_spline_lengths(coeffs_x, coeffs_y, no_interp_points=15):
    t = np.linspace(0.0, 1.0, no_interp_points)
    T = np.stack((np.ones_like(t), t, t ** 2, t ** 3))          # [4, P]
    px = coeffs_x @ T
    py = coeffs_y @ T
    return np.sum(np.sqrt(np.diff(px, axis=1) ** 2 + np.diff(py, axis=1) ** 2), axis=1)


def interp_splines(coeffs_x, coeffs_y, spline_lengths, stepsize_approx, incl_last_point=False):
    dists_cum = np.cumsum(spline_lengths)
    no_interp = math.ceil(dists_cum[-1] / stepsize_approx) + 1
    dists_interp = np.linspace(0.0, dists_cum[-1], no_interp)
    m = no_interp - 1
    path = np.zeros((m, 2))
    inds = np.zeros(m, dtype=int)
    tv = np.zeros(m)
    for i in range(m):
        j = int(np.argmax(dists_interp[i] < dists_cum))
        inds[i] = j
        tv[i] = (dists_interp[i] - (dists_cum[j - 1] if j > 0 else 0.0)) / spline_lengths[j]
        t = tv[i]
        path[i, 0] = coeffs_x[j, 0] + coeffs_x[j, 1] * t + coeffs_x[j, 2] * t ** 2 + coeffs_x[j, 3] * t ** 3
        path[i, 1] = coeffs_y[j, 0] + coeffs_y[j, 1] * t + coeffs_y[j, 2] * t ** 2 + coeffs_y[j, 3] * t ** 3
    if incl_last_point:
        raise NotImplementedError
    return path, inds, tv, dists_interp[:-1]


def create_raceline(refline, normvectors, alpha, stepsize_interp):
    """Boundary [REF main_globaltraj.py:371-376]; 9-tuple as unpacked there."""
    raceline = refline + alpha[:, None] * normvectors
    raceline_cl = np.vstack((raceline, raceline[0]))
    cx, cy, A_rl, _ = calc_splines(raceline_cl, use_dist_scaling=False)
    lengths = calc_spline_lengths(cx, cy)
    rl_interp, inds, tv, s_interp = interp_splines(cx, cy, lengths, stepsize_interp)
    s_tot = float(np.sum(lengths))
    el = np.append(np.diff(s_interp), s_tot - s_interp[-1])
    return rl_interp, A_rl, cx, cy, inds, tv, s_interp, lengths, el


def interp_track_widths(w_track, spline_inds, t_values):
    w_cl = np.vstack((w_track, w_track[0]))
    lo = w_cl[spline_inds]
    hi = w_cl[spline_inds + 1]
    return lo + (hi - lo) * t_values[:, None]


def iqp_handler(reftrack, normvectors, A, kappa_bound, w_veh, stepsize_interp, iters_min=3, curv_error_allowed=0.01,
                solver=None, trace=None):
    """Dense-faithful restatement of tph.iqp_handler.iqp_handler, boundary [REF main_globaltraj.py:273-284].

    Returns (alpha[N'], reftrack[N',4], normvectors[N',2]) of the LAST re-linearisation.  Works on a copy of
    `reftrack` (upstream aliases the caller's array on the first pass; main rebinds the name so this is unobservable).
    """
    reftrack_tmp = np.array(reftrack, dtype=np.float64)
    normvec_tmp = np.array(normvectors, dtype=np.float64)
    A_tmp = A
    it = 0
    while True:
        it += 1
        alpha, err = opt_min_curv(reftrack_tmp, normvec_tmp, A_tmp, kappa_bound, w_veh, solver=solver)
        if trace is not None:
            trace.append(dict(iter=it, n=reftrack_tmp.shape[0], curv_error_max=err, alpha=alpha.copy(),
                              reftrack=reftrack_tmp.copy(), normvec=normvec_tmp.copy()))
        if it < iters_min:
            alpha = alpha * (it * 1.0 / iters_min)
        if it >= iters_min and err <= curv_error_allowed:
            break
        refline_tmp, _, _, _, inds, tv = create_raceline(reftrack_tmp[:, :2], normvec_tmp, alpha, stepsize_interp)[:6]
        reftrack_tmp[:, 2] -= alpha
        reftrack_tmp[:, 3] += alpha
        ws = interp_track_widths(reftrack_tmp[:, 2:], inds, tv)
        reftrack_tmp = np.column_stack((refline_tmp, ws))
        refline_cl = np.vstack((reftrack_tmp[:, :2], reftrack_tmp[0, :2]))
        _, _, A_tmp, normvec_tmp = calc_splines(refline_cl, use_dist_scaling=False)
    return alpha, reftrack_tmp, normvec_tmp
