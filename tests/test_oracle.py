"""CPU tests that pin the oracle against itself and against the committed golden vectors (PARITY UNPINNED by the
reference: it has no tests; see oracle/tph_ref.py header and SURVEY.md section 8c)."""
import os

import numpy as np
import pytest

from oracle import qp_ref, tph_ref


def _circle(n=120, radius=57.3, ccw=True):
    th = np.linspace(0.0, 2 * np.pi, n, endpoint=False)
    if not ccw:
        th = -th
    xy = radius * np.column_stack((np.cos(th), np.sin(th)))
    return xy


def _prep(xy, w_r, w_l):
    path_cl = np.vstack((xy, xy[0]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    ref = np.column_stack((xy, np.full(xy.shape[0], w_r), np.full(xy.shape[0], w_l)))
    return ref, nv, A


def test_gi_dense_against_bvls_and_kkt(golden):
    for name in ("rounded_rectangle", "handling_track"):
        g = golden[name]
        path_cl = np.vstack((g["reftrack"][:, :2], g["reftrack"][0, :2]))
        _, _, A, nv = tph_ref.calc_splines(path_cl)
        assert np.max(np.abs(nv - g["normvec"])) < 1e-13
        alpha, err, I = tph_ref.opt_min_curv(g["reftrack"], nv, A, 0.12, 3.4, return_internals=True)
        assert np.max(np.abs(alpha - g["alpha"])) < 1e-10
        assert abs(err - float(g["curv_error_max"])) < 1e-12
        lo, hi = -(g["reftrack"][:, 3] - 1.7), g["reftrack"][:, 2] - 1.7
        a2 = qp_ref.solve_box_bvls(I["E"], I["k_ref"], lo, hi)
        assert np.max(np.abs(a2 - alpha)) < 1e-9          # two independent routes agree
        k = qp_ref.kkt_residuals(I["H"], I["f"], I["G"], I["h"], alpha)
        assert k["stationarity"] < 1e-10 and k["primal"] < 1e-10
        # structural identities the GPU formulation relies on
        assert np.max(np.abs(I["H"] - I["E"].T @ I["E"])) < 1e-14 * np.max(np.abs(I["H"])) * 100
        assert np.max(np.abs(I["f"] - 2.0 * I["E"].T @ I["k_ref"])) < 1e-12 * np.max(np.abs(I["f"]))


def test_factor_two_quirk_is_reproduced(golden):
    """quadprog sees 1/2 a'Ha + f'a with f = 2 E'k_ref; F_SCALE = 1 gives a different alpha."""
    g = golden["rounded_rectangle"]
    path_cl = np.vstack((g["reftrack"][:, :2], g["reftrack"][0, :2]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    H, f, E, k_ref, _ = tph_ref.assemble_dense(g["reftrack"], nv, A)
    Gm, h = tph_ref.constraints_dense(g["reftrack"], E, k_ref, 0.12, 3.4)
    a_two = qp_ref.solve_qp_gi(H, f, Gm, h)
    a_one = qp_ref.solve_qp_gi(H, 0.5 * f, Gm, h)
    assert np.max(np.abs(a_two - g["alpha"])) < 1e-10
    assert np.max(np.abs(a_one - g["alpha"])) > 1e-3


def test_known_answer_circle():
    """Circle, uniform widths: by symmetry alpha is constant; the single-shot QP moves the line inward to the bound
    (SURVEY.md section 8c known-answer tests)."""
    xy = _circle()
    ref, nv, A = _prep(xy, 5.0, 5.0)
    alpha, _ = tph_ref.opt_min_curv(ref, nv, A, 0.12, 3.4)
    assert np.max(np.abs(alpha - alpha[0])) < 1e-8
    assert abs(abs(alpha[0]) - 3.3) < 1e-9


def test_too_narrow_raises():
    xy = _circle()
    ref, nv, A = _prep(xy, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="Problem not solvable"):
        tph_ref.opt_min_curv(ref, nv, A, 0.12, 3.4)


def test_rotation_and_mirror_invariance(golden):
    g = golden["rounded_rectangle"]
    ref = g["reftrack"]
    n = ref.shape[0]
    # roll the start index
    refr = np.roll(ref, 17, axis=0)
    path_cl = np.vstack((refr[:, :2], refr[0, :2]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    ar, _ = tph_ref.opt_min_curv(refr, nv, A, 0.12, 3.4)
    assert np.max(np.abs(np.roll(ar, -17) - g["alpha"])) < 1e-8
    # mirror y, swap widths -> alpha changes sign
    refm = ref.copy()
    refm[:, 1] *= -1.0
    refm[:, [2, 3]] = refm[:, [3, 2]]
    path_cl = np.vstack((refm[:, :2], refm[0, :2]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    am, _ = tph_ref.opt_min_curv(refm, nv, A, 0.12, 3.4)
    assert np.max(np.abs(am + g["alpha"])) < 1e-8
    assert n == 105


def test_gi_reports_infeasible_and_not_pd():
    H = np.eye(2)
    f = np.zeros(2)
    G = np.array([[1.0, 0.0], [-1.0, 0.0]])
    h = np.array([-1.0, -1.0])     # x <= -1 and x >= 1
    with pytest.raises(ValueError, match="inconsistent"):
        qp_ref.solve_qp_gi(H, f, G, h)
    with pytest.raises(ValueError, match="positive definite"):
        qp_ref.solve_qp_gi(np.array([[1.0, 2.0], [2.0, 1.0]]), f, G, np.array([1.0, 1.0]))


def test_gi_dense_reproduces_the_example_of_quadprogs_own_documentation():
    """The one published known answer of the solver the reference's path ends in: the example of R quadprog's `solve.QP` help page (the Fortran
    `qpgen2` the Python package `quadprog` compiles [REF Readme.md:40,44]; also the case its own test-suite solves) --
    Dmat = I3, dvec = (0, 5, 0), Amat = [[-4, 2, 0], [-3, 1, -2], [0, 0, 1]], bvec = (-8, 2, 0) -> solution (0.4761905, 1.0476190, 2.0952381),
    value -2.380952, Lagrangian (0, 0.2380952, 2.0952381), iterations (3, 0), iact (3, 2) (1-based there).  oracle/gi_dense.c returns every
    field of that tuple, the iteration pair and the ORDER of the active set included -- its rule set is qpgen2's."""
    Amat = np.array([[-4.0, 2.0, 0.0], [-3.0, 1.0, -2.0], [0.0, 0.0, 1.0]])
    x, fval, lagr, iact, iters = qp_ref.solve_qp_quadprog_convention(np.eye(3), np.array([0.0, 5.0, 0.0]), Amat, np.array([-8.0, 2.0, 0.0]))
    assert np.max(np.abs(x - np.array([10.0, 22.0, 44.0]) / 21.0)) < 1e-15
    assert abs(fval + 50.0 / 21.0) < 1e-15
    assert np.max(np.abs(lagr - np.array([0.0, 5.0 / 21.0, 44.0 / 21.0]))) < 1e-15
    assert list(iact) == [2, 1] and list(iters) == [3, 0]
    # an equality row (meq = 1): x0 + x1 = 1 with the box x >= 0 -> the projection of (0, 5, 0) onto the plane
    C = np.array([[1.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]).T
    x, _, lagr, iact, iters = qp_ref.solve_qp_quadprog_convention(np.eye(3), np.array([0.0, 5.0, 0.0]), C, np.array([1.0, 0.0, 0.0, 0.0]), meq=1)
    assert np.max(np.abs(x - np.array([0.0, 1.0, 0.0]))) < 1e-15 and sorted(iact) == [0, 1]


def test_gi_dense_against_brute_force_on_tiny_random_qps():
    """The restated qpgen2 against an enumeration of active sets on 400 tiny strictly convex QPs (n <= 4, up to 7 inequality rows, some with an
    equality, about a third infeasible): where a KKT point exists the solver returns it (x to 1e-9, multipliers non-negative and complementary, the
    criterion value quadprog's tuple carries), where none does it reports "constraints are inconsistent" -- adds, drops, partial steps and the
    z = 0 branch all occur at these sizes (iters[1] > 0 in a good part of them)."""
    import itertools
    rng = np.random.default_rng(11)
    n_sol = n_inf = n_drop = 0
    for trial in range(400):
        n = int(rng.integers(2, 5))
        m = int(rng.integers(1, 8))
        meq = int(rng.integers(0, 2)) if m >= 2 else 0
        Mx = rng.standard_normal((n, n))
        Gm = Mx @ Mx.T + 0.1 * np.eye(n)
        a = rng.standard_normal(n) * 2.0
        C = rng.standard_normal((n, m))
        b = rng.standard_normal(m) * (1.5 if trial % 3 == 0 else 0.5)
        if trial % 7 == 0 and m >= 2:                       # an opposite pair that cannot both hold: C_1 = -C_0, b_0 + b_1 > 0
            C[:, 1] = -C[:, 0]
            b[0], b[1] = 0.5, 0.2
        # brute force: every subset of inequality rows held as equalities (with the meq equality rows), KKT conditions checked
        best = None
        ineq = list(range(meq, m))
        for r in range(0, min(len(ineq), n - meq) + 1):
            for act in itertools.combinations(ineq, r):
                rows = list(range(meq)) + list(act)
                k = len(rows)
                if k > n:
                    continue
                Ca = C[:, rows]
                K = np.block([[Gm, -Ca], [Ca.T, np.zeros((k, k))]]) if k else Gm
                rhs = np.concatenate((a, b[rows])) if k else a
                try:
                    sol = np.linalg.solve(K, rhs)
                except np.linalg.LinAlgError:
                    continue
                x, lam = sol[:n], sol[n:]
                if k and np.linalg.cond(K) > 1e10:
                    continue
                if np.all(C.T @ x - b >= -1e-9) and np.all(lam[meq:] >= -1e-9) and (meq == 0 or np.all(np.abs(C[:, :meq].T @ x - b[:meq]) < 1e-9)):
                    f = 0.5 * x @ Gm @ x - a @ x
                    if best is None or f < best[1] - 1e-12:
                        best = (x, f)
        try:
            x, fval, lagr, iact, iters = qp_ref.solve_qp_quadprog_convention(Gm, a, C, b, meq=meq)
        except ValueError as e:
            assert "inconsistent" in str(e)
            assert best is None, (trial, "the solver says inconsistent, enumeration found a KKT point")
            n_inf += 1
            continue
        assert best is not None, (trial, "the solver returned a point, enumeration found none")
        sc = 1.0 + float(np.max(np.abs(best[0])))
        assert np.max(np.abs(x - best[0])) < 1e-8 * sc and abs(fval - best[1]) < 1e-9 * (1.0 + abs(best[1])), (trial, x, best[0])
        s_ = C.T @ x - b
        assert np.all(s_[meq:] >= -1e-9 * sc) and np.all(lagr[meq:] >= 0.0) and np.max(np.abs(lagr[meq:] * s_[meq:])) < 1e-8 * sc * (1.0 + float(np.max(lagr)))
        if meq == 0:        # stationarity with quadprog's multipliers: G x - a = C lagr
            assert np.max(np.abs(Gm @ x - a - C @ lagr)) < 1e-8 * (1.0 + float(np.max(np.abs(Gm @ x - a))))
        n_sol += 1
        n_drop += 1 if iters[1] > 0 else 0
    assert n_sol > 150 and n_inf > 40 and n_drop > 20, (n_sol, n_inf, n_drop)


def test_iqp_golden_shapes(golden):
    g = golden["rounded_rectangle"]
    assert list(g["iqp_n"]) == [105, 104, 103]
    assert g["iqp_curv_err"][-1] <= 0.01 < g["iqp_curv_err"][-2]
    assert g["iqp_alpha"].shape[0] == g["iqp_reftrack"].shape[0] == g["iqp_normvec"].shape[0] == 103


def test_shortest_path_oracle_encodes_the_polygon_length(golden):
    """Row f-4: the restated H, f of tph.opt_shortest_path are the quadratic form of the sum of squared segment lengths
    (checked at random shifts), H is the cyclic tridiagonal the device kernel writes entry by entry, and the committed
    alphas are the oracle's own (KKT certificate) and shorten the polygon."""
    g = golden["handling_track"]
    ref, nv = g["reftrack"], g["normvec"]
    n = ref.shape[0]
    H, f, G, h = tph_ref.shortest_path_dense(ref, nv, 3.4)
    rng = np.random.default_rng(5)
    c0 = tph_ref.path_length_sq(ref, nv, np.zeros(n))
    for _ in range(3):
        a = rng.uniform(-2.0, 2.0, n)
        assert abs(0.5 * a @ H @ a + f @ a + c0 - tph_ref.path_length_sq(ref, nv, a)) < 1e-9 * c0
    i = np.arange(n)
    T = np.zeros((n, n))
    T[i, i] = 4.0 * np.sum(nv * nv, axis=1)
    off = -2.0 * np.sum(nv * np.roll(nv, -1, axis=0), axis=1)
    T[i, (i + 1) % n] = off
    T[(i + 1) % n, i] = off
    assert np.max(np.abs(H - T)) < 1e-14
    p = ref[:, :2]
    assert np.max(np.abs(f - 2.0 * np.sum(nv * (2 * p - np.roll(p, 1, axis=0) - np.roll(p, -1, axis=0)), axis=1))) < 1e-11
    # narrow stretches are clipped to 1 mm, not rejected
    _, _, _, h_wide = tph_ref.shortest_path_dense(ref, nv, 50.0)
    assert np.all(h_wide == 0.001)
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "shortest_path.npz"))
    alpha = tph_ref.opt_shortest_path(ref, nv, float(z["w_veh"]))
    assert np.max(np.abs(alpha - z["handling_track_alpha"])) < 1e-10
    kkt = qp_ref.kkt_residuals(H, f, G, h, alpha)
    assert kkt["stationarity"] < 1e-12 and kkt["primal"] < 1e-12
    assert tph_ref.path_length_sq(ref, nv, alpha) < c0


def test_cpu_b_banded_solver_matches_dense_oracle(golden):
    """oracle/banded_qp.c ("CPU-B": cyclic-tridiagonal assembly, E band of half width 36, bordered-band interior point + active
    set + refinement through E, scalar C) against the dense-faithful oracle's committed alpha on the reference's two long
    tracks and on the N = 2000 oval -- three routes now agree there (dense Goldfarb-Idnani, trust-region-reflective least
    squares, banded interior point / active set); and its status codes."""
    import os
    from oracle import banded_ref
    cases = [golden["modena_2019"], golden["berlin_2018"]]
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "oval_n2000.npz"))
    cases.append({k: z[k] for k in z.files})
    for g in cases:
        a, c, st, it, used = banded_ref.solve_batch(g["reftrack"][None], g["normvec"][None], g["scaling"][None], 0.12, 3.4)
        assert st[0] == 0 and used == 1
        assert np.max(np.abs(a[0] - g["alpha"])) < 1e-8
        assert abs(c[0] - float(g["curv_error_max"])) < 1e-12
        assert 5 <= it[0, 0] <= 40 and 1 <= it[0, 1] <= 20
    # a batch over threads returns what the single calls return, bitwise
    g = cases[0]
    refs = np.stack([g["reftrack"]] * 3)
    refs[1, :, 2:] += 0.3
    refs[2, :, 2:] = 1.0                                   # w_r + w_l < w_veh: infeasible widths
    a3, c3, st3, _, used = banded_ref.solve_batch(refs, np.stack([g["normvec"]] * 3), np.stack([g["scaling"]] * 3), 0.12, 3.4, nthreads=3)
    a0, _, _, _, _ = banded_ref.solve_batch(refs[:1], g["normvec"][None], g["scaling"][None], 0.12, 3.4)
    assert list(st3) == [0, 0, 1] and np.array_equal(a3[0], a0[0]) and not np.array_equal(a3[0], a3[1])
    # a ring too short for its bordered band is refused, not mis-solved
    h = golden["handling_track"]
    assert banded_ref.solve_batch(h["reftrack"][None], h["normvec"][None], h["scaling"][None], 0.12, 3.4)[2][0] == 4
    # curvature rows are checked: a bound below the curvature at the box optimum is reported (status 6)
    assert banded_ref.solve_batch(g["reftrack"][None], g["normvec"][None], g["scaling"][None], 0.05, 3.4)[2][0] == 6


def test_round4_goldens_are_consistent_with_cpu_b():
    """The dense-oracle fixtures of round 4 (N = 2000, scripts/make_golden_r4.py) against a THIRD route at full size: CPU-B
    (oracle/banded_qp.c -- its own cyclic-tridiagonal assembly, a truncated band of E, bordered-band interior point + active set), which
    shares no code with oracle/tph_ref.py + gi_dense.c.  Also: the stored KKT stationarity of every fixture (tests/golden/SUMMARY_r4.json)."""
    import json
    import os
    from conftest import GOLDEN_DIR, load_golden
    from oracle import banded_ref
    summ = json.load(open(os.path.join(GOLDEN_DIR, "SUMMARY_r4.json")))
    names = ("oval_n2000_w3", "oval_n2000_w7", "oval_n2000_w11", "oval_n2000_c13", "oval_n2000_c21", "iqp_pass2_oval5", "iqp_pass3_oval9")
    for name in names + ("oval_n2000_kappa",):
        assert summ[name]["kkt_stationarity"] < 1e-10, name
    assert summ["oval_n2000_kappa"]["n_active_kappa"] >= 5
    for name in names:
        g = load_golden(name)
        n = g["reftrack"].shape[0]
        sc = g["scaling"] if "scaling" in g else np.ones(n)
        a, c, st, _, _ = banded_ref.solve_batch(g["reftrack"][None], g["normvec"][None], sc[None], float(g["kappa_bound"]), float(g["w_veh"]))
        assert st[0] == 0, name
        assert np.max(np.abs(a[0] - g["alpha"])) < 2e-8, (name, float(np.max(np.abs(a[0] - g["alpha"]))))
        assert abs(c[0] - float(g["curv_error_max"])) < 1e-9, name


def test_round5_goldens_above_2048_waypoints_are_consistent_with_cpu_b():
    """above 2048 waypoints the engine's long-ring route had only CPU-B (the same author's interior point + block pivoting
    design) to be compared with, and CPU-B was tied to the dense oracle at N <= 2003 only.  scripts/make_golden_r5.py puts the dense oracle
    (10 400 x 10 400 inverse, dense Goldfarb-Idnani with all 4N rows) at N = 2100 and N = 2600: CPU-B agrees with it there too, the stored KKT
    stationarity is at rounding level, and the curvature-bound fixture has curvature rows in its working set."""
    import json
    import os
    from conftest import GOLDEN_DIR, load_golden
    from oracle import banded_ref
    summ = json.load(open(os.path.join(GOLDEN_DIR, "SUMMARY_r5.json")))
    for name in ("oval_n2100", "oval_n2600", "oval_n2600_kappa"):
        assert summ[name]["kkt_stationarity"] < 1e-10, name
    assert summ["oval_n2600_kappa"]["n_active_kappa"] >= 3
    for name in ("oval_n2100", "oval_n2600"):
        g = load_golden(name)
        assert g["reftrack"].shape[0] > 2048
        a, c, st, _, _ = banded_ref.solve_batch(g["reftrack"][None], g["normvec"][None], g["scaling"][None], float(g["kappa_bound"]), float(g["w_veh"]))
        assert st[0] == 0, name
        assert np.max(np.abs(a[0] - g["alpha"])) < 2e-8, (name, float(np.max(np.abs(a[0] - g["alpha"]))))
        assert abs(c[0] - float(g["curv_error_max"])) < 1e-9, name
    # the curvature-tight fuzz fixture: every stored solution is feasible for its own (tight) bound and box
    z = np.load(os.path.join(GOLDEN_DIR, "kappa_tight_fuzz.npz"))
    off = z["offsets"]
    assert len(off) - 1 >= 200 and int(np.sum(z["status_ref"] == 5)) >= 10 and int(np.max(z["n_active_kappa"])) > 120
    for k in (0, 17, 95, 200):
        if z["status_ref"][k] != 0:
            continue
        ref, al = z["reftrack"][off[k]:off[k + 1]], z["alpha"][off[k]:off[k + 1]]
        w = float(z["w_veh"][k])
        assert np.all(al <= ref[:, 2] - w / 2 + 1e-9) and np.all(al >= -(ref[:, 3] - w / 2) - 1e-9)
