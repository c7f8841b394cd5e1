"""The engine's Goldfarb-Idnani path (csrc/mcq_gi.inc) on the MI355X, through the C ABI: quadprog's algorithm [REF requirements.txt:3 via
tph.opt_min_curv, main_globaltraj.py:264-271; params/racecar.ini:49 curvlim] as the fallback of the block-pivoting phase -- VERDICT r4 item 1: no
feasible strictly convex QP may end in MCQ_ITER_CAP / MCQ_KAPPA_ACTIVE, whatever the rounding sequence -- and on its own
(mcq_opts.algorithm = MCQ_ALG_GI).  Oracle: dense Goldfarb-Idnani with all 4N rows (oracle/gi_dense.c), live for the stadiums, committed
vectors (scripts/make_golden_r5.py) for the curvature-tight fuzz."""
import os

import numpy as np
import pytest

from global_racetrajectory_optimization_amd import engine, synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALPHA_TOL = 1e-6        # north_star's fp64 tolerance against quadprog


def _stadium_cases():
    from oracle import qp_ref, tph_ref
    from test_emu_kernels import stadium_problem
    probs, want = [], []
    for n in (360, 720):
        ref, nv, A, sc, kb = stadium_problem(n, 0.0223)
        info = {}
        a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, kb, 2.0, solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
        probs.append(dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=kb, w_veh=2.0))
        want.append((a_ref, err_ref, int(np.sum(info["lagr"][2 * n:] > 0))))
    return probs, want


@pytest.fixture(scope="module")
def stadiums():
    return _stadium_cases()


def test_gi_mode_reference_tracks_and_stadiums(gpu_engine, golden, stadiums):
    """Every problem through the Goldfarb-Idnani path alone: the reference's four tracks against their dense-oracle goldens -- the same
    vertex in the same number of steps as the dense Goldfarb-Idnani took (both add the most violated row and drop by the same ratio test) --
    and the stadiums with 134 / 270 active curvature rows (no limit on the working set: MCQ_KMAX and the overflow slots are the block-pivoting
    phase's)."""
    names = ("rounded_rectangle", "handling_track", "modena_2019", "berlin_2018")
    probs = [dict(reftrack=golden[t]["reftrack"], normvec=golden[t]["normvec"], scaling=golden[t]["scaling"], kappa_bound=0.12, w_veh=3.4)
             for t in names]
    sp, want = stadiums
    al, curv, st, info = gpu_engine.solve_batch(probs + sp, algorithm=engine.ALG_GI)
    assert np.all(st == 0), st
    for k, t in enumerate(names):
        assert np.max(np.abs(al[k] - golden[t]["alpha"])) < 2e-8, (t, float(np.max(np.abs(al[k] - golden[t]["alpha"]))))
        assert abs(curv[k] - float(golden[t]["curv_error_max"])) < 1e-9
        assert info[k]["second_attempt"] & 4 and info[k]["gi_iters"] > 0 and info[k]["ipm_iters"] == 0
    for k, (a_ref, err_ref, nk) in enumerate(want):
        i = info[len(names) + k]
        assert i["n_active_kappa"] == nk, (i["n_active_kappa"], nk)
        assert np.max(np.abs(al[len(names) + k] - a_ref)) < ALPHA_TOL and abs(curv[len(names) + k] - err_ref) < 1e-9
    print("GI mode: steps %s, ms per problem %s, polish rejected %s, max |alpha - oracle| %s" % (
        [i["gi_iters"] for i in info], ["%.1f" % (i["ticks"][3] / 1e5) for i in info], [bool(i["second_attempt"] & 8) for i in info],
        ["%.1e" % float(np.max(np.abs(al[k] - (golden[names[k]]["alpha"] if k < 4 else want[k - 4][0])))) for k in range(len(al))]))
    # against the default path on the same problems
    al0, curv0, st0, _ = gpu_engine.solve_batch(probs)
    assert np.all(st0 == 0) and max(float(np.max(np.abs(a - b))) for a, b in zip(al0, al[:4])) < 1e-9


def test_arithmetic_variants_on_the_stadium(stadiums):
    """VERDICT r4 item 1(i).  Round 4's block-pivoting phase solved the 720-point stadium (270 adjacent active curvature rows) by the grace of
    one rounding sequence: a one-ulp perturbation of every solve's separator values turned status 0 into 6 / 3 (docs/NOTEBOOK.md R4.6).  The
    SAME sources built four more ways (__graft_entry__.VARIANTS: that perturbation, interprocedural register allocation on, no compiler-formed
    fused multiply-adds, fp64 records throughout the interior point) must return the dense oracle's vertex on both stadiums -- by block
    pivoting where it settles, through the Goldfarb-Idnani path where it does not."""
    import __graft_entry__ as ge
    probs, want = stadiums
    report = {}
    for name in ge.VARIANTS:
        path = ge.variant_path(name)
        assert os.path.exists(path), "arithmetic variant '%s' was not built (__graft_entry__.build_variants)" % name
        eng = engine.Engine(0, lib_path=path)
        try:
            al, curv, st, info = eng.solve_batch(probs)
        finally:
            eng.close()
        assert np.all(st == 0), (name, st)
        for k, (a_ref, err_ref, nk) in enumerate(want):
            d = float(np.max(np.abs(al[k] - a_ref)))
            assert d < ALPHA_TOL, (name, k, d)
            assert abs(curv[k] - err_ref) < 1e-9 and info[k]["n_active_kappa"] == nk, (name, k)
        report[name] = [(i["as_iters"], i["gi_iters"]) for i in info]
    print("stadium 360 / 720, (block-pivoting rounds, Goldfarb-Idnani steps) per build:", report)


def test_curvature_tight_fuzz_against_dense_gi(gpu_engine):
    """220 problems with kappa_bound between 0.6 x and 1.0 x the curvature maximum of their box optimum (stadiums,
    star-shaped rings, the reference's own tracks; tests/golden/kappa_tight_fuzz.npz, scripts/make_golden_r5.py) in one ragged launch.
    Status 0 and the dense oracle's vertex for every problem the dense Goldfarb-Idnani solves; MCQ_KAPPA_INFEASIBLE exactly where it reports
    "constraints are inconsistent"; no other status."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "kappa_tight_fuzz.npz"))
    off = z["offsets"]
    nprob = len(off) - 1
    assert nprob >= 200
    probs = [dict(reftrack=z["reftrack"][off[k]:off[k + 1]], normvec=z["normvec"][off[k]:off[k + 1]], scaling=z["scaling"][off[k]:off[k + 1]],
                  kappa_bound=float(z["kappa_bound"][k]), w_veh=float(z["w_veh"][k])) for k in range(nprob)]
    al, curv, st, info = gpu_engine.solve_batch(probs)
    st_ref = z["status_ref"]
    assert np.array_equal(np.asarray(st) == 0, st_ref == 0), [(k, int(st[k]), int(st_ref[k])) for k in range(nprob) if (st[k] == 0) != (st_ref[k] == 0)]
    assert set(np.unique(st)) <= {0, engine.STATUS_KAPPA_INFEASIBLE}, np.unique(st)
    worst = 0.0
    for k in range(nprob):
        if st_ref[k] != 0:
            continue
        d = float(np.max(np.abs(al[k] - z["alpha"][off[k]:off[k + 1]])))
        worst = max(worst, d)
        assert d < ALPHA_TOL, (k, d)
        assert abs(curv[k] - float(z["curv_error_max"][k])) < 1e-8, k
        assert info[k]["n_active_kappa"] == int(z["n_active_kappa"][k]), (k, info[k]["n_active_kappa"], int(z["n_active_kappa"][k]))
    ran = [k for k in range(nprob) if info[k]["second_attempt"] & 4]
    print("curvature-tight fuzz: %d problems (%d inconsistent), max |alpha - dense GI| %.2e m, Goldfarb-Idnani path ran for %d; for feasible "
          "problems (index, status the block-pivoting phase had left, active curvature rows, steps): %s" % (
              nprob, int(np.sum(st_ref != 0)), worst, len(ran),
              [(k, (info[k]["second_attempt"] >> 4) & 15, info[k]["n_active_kappa"], info[k]["gi_iters"]) for k in ran if st_ref[k] == 0]))
    # an exchange with curvature rows hands over where the single-pivot rule would begin, not at the cap of 60 rounds (docs/NOTEBOOK.md R5.6)
    for k in ran:
        if st_ref[k] == 0 and (info[k]["second_attempt"] >> 4) & 15 == engine.STATUS_ITER_CAP:
            assert info[k]["as_iters"] <= 24, (k, info[k])
    # the same set through the Goldfarb-Idnani path alone
    al2, curv2, st2, info2 = gpu_engine.solve_batch(probs, algorithm=engine.ALG_GI)
    assert np.array_equal(np.asarray(st2), np.where(st_ref == 0, 0, engine.STATUS_KAPPA_INFEASIBLE))
    w2 = max(float(np.max(np.abs(al2[k] - z["alpha"][off[k]:off[k + 1]]))) for k in range(nprob) if st_ref[k] == 0)
    assert w2 < ALPHA_TOL, w2
    print("the same through the Goldfarb-Idnani path alone: max |alpha - dense GI| %.2e m, steps mean %.0f max %d" % (
        w2, float(np.mean([i["gi_iters"] for i in info2])), max(i["gi_iters"] for i in info2)))


def test_long_rings_against_dense_goldens(gpu_engine):
    """the dense oracle ABOVE 2048 waypoints (scripts/make_golden_r5.py: N = 2100 and 2600 through the dense
    10 400 x 10 400 inverse and the dense Goldfarb-Idnani with all 4N rows; one with the curvature bound active; tph.opt_shortest_path's QP
    at N = 2100).  This is the range of the long-ring route -- tridiagonal sweeps on workspace vectors, the general interior point -- that was
    silently wrong for a whole round (2049 .. 2208 waypoints, metres of error) while every test was green, because only CPU-B looked at it.
    Reference behaviour: any N [REF helper_funcs_glob/src/prep_track.py:39-51 sets N from stepsize_reg]."""
    from conftest import load_golden
    names = ("oval_n2100", "oval_n2600", "oval_n2600_kappa")
    gs = [load_golden(t) for t in names]
    probs = [dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=float(g["kappa_bound"]), w_veh=float(g["w_veh"]))
             for g in gs]
    al, curv, st, info = gpu_engine.solve_batch(probs)
    assert np.all(st == 0), st
    d = [float(np.max(np.abs(al[k] - gs[k]["alpha"]))) for k in range(3)]
    assert max(d) < ALPHA_TOL, d
    assert max(abs(curv[k] - float(gs[k]["curv_error_max"])) for k in range(3)) < 1e-9
    assert info[2]["n_active_kappa"] >= 3 and abs(info[2]["kappa_max"] - float(gs[2]["kappa_bound"])) < 1e-10
    # the same three through the Goldfarb-Idnani path alone (its solves and E / E' on the long-ring route)
    al2, curv2, st2, info2 = gpu_engine.solve_batch(probs, algorithm=engine.ALG_GI)
    assert np.all(st2 == 0), st2
    d2 = [float(np.max(np.abs(al2[k] - gs[k]["alpha"]))) for k in range(3)]
    assert max(d2) < ALPHA_TOL, d2
    g = load_golden("shortest_path_n2100")
    al3, _, st3, _ = gpu_engine.solve_batch([dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=None, kappa_bound=1.0, w_veh=float(g["w_veh"]))],
                                            objective=engine.OBJ_SHORTEST_PATH)
    d3 = float(np.max(np.abs(al3[0] - g["alpha"])))
    assert st3[0] == 0 and d3 < ALPHA_TOL, (st3[0], d3)
    print("rings above 2048 waypoints, max |alpha - dense oracle|: default path %s, Goldfarb-Idnani path %s (steps %s), shortest path %.1e" % (
        ["%.1e" % v for v in d], ["%.1e" % v for v in d2], [i["gi_iters"] for i in info2], d3))


def test_gi_mode_on_every_full_size_golden(gpu_engine):
    """Every dense-oracle fixture of the bench size through the Goldfarb-Idnani path alone (mcq_opts.algorithm = MCQ_ALG_GI): twelve first passes
    of N = 2000 ovals, the QPs of IQP second / third passes (unit scalings, dozens of bounds touched with multipliers down to 1e-7 of the gradient
    scale: the instances that made block pivoting need a second attempt), the ring with the curvature bound active at this size, Berlin at
    N = 333 / 776 and Modena.  One ragged launch of 20 problems on 8 slots; the same vertex as the default path, the oracle's to 1e-6 m."""
    from conftest import load_golden
    names = ["oval_n2000", "oval_n2000_w1", "oval_n2000_w2", "oval_n2000_w3", "oval_n2000_w7", "oval_n2000_w11", "oval_n2000_c5", "oval_n2000_c9",
             "oval_n2000_c13", "oval_n2000_c21", "oval_n2000_kappa", "iqp_pass2_oval5", "iqp_pass3_oval3", "iqp_pass3_oval9", "iqp_pass3_oval629",
             "berlin_2018_n333", "berlin_2018", "modena_2019"]
    gs = [load_golden(t) for t in names]
    probs = [dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"] if "scaling" in g else None,
                  kappa_bound=float(g["kappa_bound"]) if "kappa_bound" in g else 0.12, w_veh=float(g["w_veh"]) if "w_veh" in g else 3.4) for g in gs]
    al, curv, st, info = gpu_engine.solve_batch(probs, algorithm=engine.ALG_GI)
    assert np.all(st == 0), dict(zip(names, st))
    al0, curv0, st0, info0 = gpu_engine.solve_batch(probs)
    assert np.all(st0 == 0)
    d = [float(np.max(np.abs(al[k] - gs[k]["alpha"]))) for k in range(len(names))]
    d0 = [float(np.max(np.abs(al[k] - al0[k]))) for k in range(len(names))]
    assert max(d) < ALPHA_TOL and max(d0) < ALPHA_TOL, (dict(zip(names, d)), dict(zip(names, d0)))
    for k, g in enumerate(gs):
        assert abs(curv[k] - float(g["curv_error_max"])) < 1e-8, names[k]
        assert info[k]["n_active_box"] == info0[k]["n_active_box"] and info[k]["n_active_kappa"] == info0[k]["n_active_kappa"], names[k]
    print("GI mode on %d full-size fixtures: steps %s, ms %s, polish rejected %d, max |alpha - oracle| %.1e, max |alpha - default path| %.1e" % (
        len(names), [i["gi_iters"] for i in info], ["%.0f" % (i["ticks"][3] / 1e5) for i in info], sum(1 for i in info if i["second_attempt"] & 8),
        max(d), max(d0)))


def test_gi_mode_stays_on_one_stream_in_the_pipelined_entries(gpu_engine):
    """with mcq_opts.algorithm = MCQ_ALG_GI the second compute stream's workspace held half the fallback's few slots, so odd steps
    of mcq_solve_host_pipelined / mcq_solve_device_stream ran ~32 problems at a time.  Round 6: MCQ_ALG_GI keeps these entries on ONE compute
    stream (its per-workgroup slots belong to the first workspace).  Three steps of 96 rings through the mode: every step bitwise the default
    path's alpha, and no step an order of magnitude slower than the others."""
    import time
    bsz, n, steps = 96, 400, 3
    ref, nv, sc = synthetic.oval_batch(bsz, n=n)
    refs = [ref.copy() for _ in range(steps)]
    for k in range(steps):
        refs[k][:, :, 2:] += 0.01 * k                    # (every step its own widths)
    outs = [np.empty((bsz, n)) for _ in range(steps)]
    t0 = time.perf_counter()
    curv, st = gpu_engine.solve_host_pipelined(refs, [nv] * steps, [sc] * steps, 0.12, 3.4, outs, algorithm=engine.ALG_GI)
    t_gi = time.perf_counter() - t0
    assert np.all(st == 0)
    for k in range(steps):
        a_def, _, st_def, _ = gpu_engine.solve_host(refs[k], nv, sc, 0.12, 3.4)
        assert np.all(st_def == 0) and np.array_equal(outs[k], a_def), k
    assert t_gi < 5.0, t_gi
