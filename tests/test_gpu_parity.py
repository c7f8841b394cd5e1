"""GPU parity tests: the real libmcq.so (hand-written HIP, gfx950) through the C ABI against the committed golden
vectors and the CPU oracle.  Tolerance (north_star): |alpha_gpu - alpha_oracle| <= 1e-6 m in fp64; observed ~1e-8."""
import numpy as np
import pytest

from global_racetrajectory_optimization_amd import engine, synthetic
from global_racetrajectory_optimization_amd import trajectory_planning_helpers as tph

pytestmark = pytest.mark.gpu

ALPHA_TOL = 1e-6        # metres, stated fp64 tolerance of BASELINE.json's north_star
CURV_TOL = 1e-9


def _problem(g):
    return dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=float(g["kappa_bound"]),
                w_veh=float(g["w_veh"]))


def test_reference_tracks_match_golden(gpu_engine, golden):
    names = list(golden)
    al, curv, st, info = gpu_engine.solve_batch([_problem(golden[k]) for k in names])
    for k, name in enumerate(names):
        g = golden[name]
        assert st[k] == 0, (name, st[k])
        assert np.max(np.abs(al[k] - g["alpha"])) < ALPHA_TOL, name
        assert abs(curv[k] - float(g["curv_error_max"])) < CURV_TOL, name
        assert info[k]["kkt_res"] < 1e-9
        lo, hi = -(g["reftrack"][:, 3] - 1.7), g["reftrack"][:, 2] - 1.7
        assert np.all(al[k] >= lo - 1e-12) and np.all(al[k] <= hi + 1e-12)


def test_berlin_n333(gpu_engine):
    """BASELINE config 2 at its second size, 'N ~ 330': the reference's own preprocessing of inputs/tracks/berlin_2018.csv with
    stepsize_reg = 7.0 m (tph.spline_approximation, FITPACK) -- NOT a subsampling of the N = 776 ring -- committed with the dense
    oracle's alpha as tests/golden/berlin_2018_n333.npz (scripts/make_golden_r3.py; second route TRF 1.3e-10 m, KKT 7e-15).
    Against the fixture, against the live dense oracle on the fixture's rows, and through the drop-in function."""
    from conftest import load_golden
    from oracle import tph_ref
    g = load_golden("berlin_2018_n333")
    ref = g["reftrack"]
    assert ref.shape == (333, 4)
    al, curv, st, info = gpu_engine.solve_batch([dict(reftrack=ref, normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12, w_veh=3.4)])
    assert st[0] == 0 and info[0]["kkt_res"] < 1e-9
    assert np.max(np.abs(al[0] - g["alpha"])) < ALPHA_TOL
    assert abs(curv[0] - float(g["curv_error_max"])) < CURV_TOL
    path_cl = np.vstack((ref[:, :2], ref[0, :2]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    assert np.max(np.abs(nv - g["normvec"])) < 1e-12
    a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, 0.12, 3.4)
    a, err = tph.opt_min_curv.opt_min_curv(ref, nv, A, 0.12, 3.4)
    assert np.max(np.abs(a - a_ref)) < ALPHA_TOL and np.max(np.abs(a_ref - g["alpha"])) < 1e-9
    assert abs(err - err_ref) < CURV_TOL


def test_iqp_handler_reference_default_flow_berlin_modena(gpu_engine, golden):
    """The reference's DEFAULT flow [REF main_globaltraj.py:273-284; params/racecar.ini:72-74: iters_min 3, curv_error_allowed
    0.01, stepsize_interp = stepsize_reg = 3.0] on its shipped track (Berlin, N = 776) and on Modena (N = 663): the END STATE of
    the whole iterated re-linearisation against the dense oracle's chain (tests/golden/*_iqp.npz: dense 4N x 4N re-spline and
    dense Goldfarb-Idnani with all 4N rows every pass, scripts/make_golden_r3.py) -- through the drop-in iqp_handler (one engine
    call, warm-started passes), the device-resident driver cold, and the host-glue driver; per-pass ring sizes and curvature
    errors against the oracle's trace."""
    from conftest import load_golden
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler as iq
    for name in ("berlin_2018", "modena_2019"):
        g, q = golden[name], load_golden(name + "_iqp")
        A = tph.calc_splines.build_les_matrix(g["reftrack"].shape[0], g["scaling"])
        outs = [tph.iqp_handler.iqp_handler(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], A=A, kappa_bound=0.12, w_veh=3.4,
                                            print_debug=False, plot_debug=False, stepsize_interp=3.0, iters_min=3,
                                            curv_error_allowed=0.01)]
        trk = [dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"])]
        for kw in (dict(device_resident=True, warm_start=False), dict(device_resident=False)):
            stt = {}
            outs.append(iq.iqp_handler_batch(trk, 0.12, 3.4, 3.0, 3, 0.01, engine=gpu_engine, stats=stt, **kw)[0])
            assert stt["rounds"] == len(q["iqp_n"])
        for a, ref_out, nv_out in outs:
            assert a.shape == q["iqp_alpha"].shape == (int(q["iqp_n"][-1]),), name
            assert np.max(np.abs(a - q["iqp_alpha"])) < ALPHA_TOL, (name, float(np.max(np.abs(a - q["iqp_alpha"]))))
            assert np.max(np.abs(ref_out - q["iqp_reftrack"])) < 1e-6, name
            assert np.max(np.abs(nv_out - q["iqp_normvec"])) < 1e-8, name
        # the trace the engine keeps per round (what print_debug prints) against the oracle's
        res = gpu_engine.iqp_batch(trk, 0.12, 3.4, 3.0, iters_min=3, curv_error_allowed=0.01)
        k = int(res["rounds"][0])
        assert k == len(q["iqp_curv_err"])
        assert np.max(np.abs(res["curv_trace"][0, :k] - q["iqp_curv_err"])) < 1e-8, name


def test_drop_in_opt_min_curv_signature(golden):
    g = golden["handling_track"]
    ref = g["reftrack"]
    path_cl = np.vstack((ref[:, :2], ref[0, :2]))
    _, _, A, nv = tph.calc_splines.calc_splines(path=path_cl)
    out = tph.opt_min_curv.opt_min_curv(reftrack=ref, normvectors=nv, A=A, kappa_bound=0.12, w_veh=3.4,
                                        print_debug=False, plot_debug=False)
    assert np.max(np.abs(out[0] - g["alpha"])) < ALPHA_TOL
    assert abs(out[1] - float(g["curv_error_max"])) < CURV_TOL


def test_iqp_handler_matches_golden(golden):
    for name in ("rounded_rectangle", "handling_track"):
        g = golden[name]
        ref = g["reftrack"].copy()
        path_cl = np.vstack((ref[:, :2], ref[0, :2]))
        _, _, A, nv = tph.calc_splines.calc_splines(path=path_cl)
        a, ref_out, nv_out = tph.iqp_handler.iqp_handler(reftrack=ref, normvectors=nv, A=A, kappa_bound=0.12, w_veh=3.4,
                                                         print_debug=False, plot_debug=False, stepsize_interp=3.0,
                                                         iters_min=3, curv_error_allowed=0.01)
        assert a.shape == g["iqp_alpha"].shape
        assert np.max(np.abs(a - g["iqp_alpha"])) < ALPHA_TOL
        assert np.max(np.abs(ref_out - g["iqp_reftrack"])) < 1e-6
        assert np.max(np.abs(nv_out - g["iqp_normvec"])) < 1e-8


def test_errors_match_reference_exceptions(golden):
    g = golden["rounded_rectangle"]
    ref = g["reftrack"].copy()
    ref[10, 2:] = 1.0
    path_cl = np.vstack((ref[:, :2], ref[0, :2]))
    _, _, A, nv = tph.calc_splines.calc_splines(path=path_cl)
    with pytest.raises(RuntimeError, match="Problem not solvable"):
        tph.opt_min_curv.opt_min_curv(ref, nv, A, 0.12, 3.4)
    with pytest.raises(RuntimeError, match="same as normvectors"):
        tph.opt_min_curv.opt_min_curv(ref, nv[:-1], A, 0.12, 3.4)
    with pytest.raises(RuntimeError, match="wrong dimensions"):
        tph.opt_min_curv.opt_min_curv(ref, nv, A[:-4, :-4], 0.12, 3.4)


def test_full_size_oval_properties(gpu_engine):
    """BASELINE config 3 size (N = 2000): size-independent properties instead of a dense oracle run --
    KKT certificate from the engine, feasibility, batch-order independence, start-index rotation, mirror symmetry."""
    ref, nv, sc = synthetic.oval_batch(6, n=2000)
    probs = [dict(reftrack=ref[b], normvec=nv[b], scaling=sc[b], kappa_bound=0.12, w_veh=3.4) for b in range(6)]
    al, curv, st, info = gpu_engine.solve_batch(probs)
    assert np.all(st == 0)
    for b in range(6):
        lo, hi = -(ref[b, :, 3] - 1.7), ref[b, :, 2] - 1.7
        assert np.all(al[b] >= lo - 1e-12) and np.all(al[b] <= hi + 1e-12)
        assert info[b]["kkt_res"] < 1e-9 and info[b]["kappa_max"] < 0.12
        assert 0 < info[b]["n_active_box"] < 2000
    # batch order independence (bitwise: one workgroup per problem, no cross-problem reduction)
    al2, _, _, _ = gpu_engine.solve_batch(probs[::-1])
    for b in range(6):
        assert np.array_equal(al[b], al2[5 - b])
    # rotation of the start index
    r = 321
    pr = dict(reftrack=np.roll(ref[0], r, axis=0), normvec=np.roll(nv[0], r, axis=0), scaling=np.roll(sc[0], r),
              kappa_bound=0.12, w_veh=3.4)
    # mirror: flip y, swap widths, normals (n_x, n_y) -> (-n_x, n_y)  => alpha -> -alpha
    refm = ref[0].copy()
    refm[:, 1] *= -1.0
    refm[:, [2, 3]] = refm[:, [3, 2]]
    nvm = nv[0].copy()
    nvm[:, 0] *= -1.0
    pm = dict(reftrack=refm, normvec=nvm, scaling=sc[0], kappa_bound=0.12, w_veh=3.4)
    al3, _, st3, _ = gpu_engine.solve_batch([pr, pm])
    assert np.all(st3 == 0)
    assert np.max(np.abs(np.roll(al3[0], -r) - al[0])) < ALPHA_TOL
    assert np.max(np.abs(al3[1] + al[0])) < ALPHA_TOL


def test_full_size_similarity_invariances(gpu_engine):
    """BASELINE's full size (1024 x N = 2000, one launch) through three more size-independent properties of the QP (round 6): it sees the track only
    through differences of neighbouring waypoints and through lengths, so (i) moving the whole track by kilometres leaves alpha where it is, (ii)
    rotating it by an arbitrary angle (normals with it) too, and (iii) scaling every length by c -- waypoints, widths, vehicle width -- with the
    curvature bound scaled by 1 / c scales alpha by c and the curvature error by 1 / c.  Every problem of the batch, against its own unmoved solve;
    tolerance 1e-7 m (the moved / rotated coordinates differ from the originals in their last bits: what is checked is that kilometre-sized
    offsets cost no more than that)."""
    bsz, n = 1024, 2000
    ref, nv, sc = synthetic.oval_batch(bsz, n=n)
    p_al = np.empty((bsz, n))
    al, curv, st, _ = gpu_engine.solve_host(ref, nv, sc, 0.12, 3.4, alpha_out=p_al)
    al = al.copy()
    assert np.all(st == 0)
    # (i) translation
    moved = ref.copy()
    moved[:, :, 0] += 12345.678
    moved[:, :, 1] -= 3210.987
    al1, curv1, st1, _ = gpu_engine.solve_host(moved, nv, sc, 0.12, 3.4)
    assert np.all(st1 == 0) and np.max(np.abs(al1 - al)) < 1e-7 and np.max(np.abs(curv1 - curv)) < 1e-9
    # (ii) rotation by 0.7 rad about the origin
    c_, s_ = np.cos(0.7), np.sin(0.7)
    rot = ref.copy()
    rot[:, :, 0] = c_ * ref[:, :, 0] - s_ * ref[:, :, 1]
    rot[:, :, 1] = s_ * ref[:, :, 0] + c_ * ref[:, :, 1]
    nvr = np.stack((c_ * nv[:, :, 0] - s_ * nv[:, :, 1], s_ * nv[:, :, 0] + c_ * nv[:, :, 1]), axis=2)
    al2, curv2, st2, _ = gpu_engine.solve_host(rot, nvr, sc, 0.12, 3.4)
    assert np.all(st2 == 0) and np.max(np.abs(al2 - al)) < 1e-7 and np.max(np.abs(curv2 - curv)) < 1e-9
    # (iii) similarity: lengths x 2.5, curvature bound / 2.5
    k = 2.5
    al3, curv3, st3, _ = gpu_engine.solve_host(ref * k, nv, sc, 0.12 / k, 3.4 * k)
    assert np.all(st3 == 0) and np.max(np.abs(al3 / k - al)) < 1e-7 and np.max(np.abs(curv3 * k - curv)) < 1e-9
    print("1024 x N = 2000: max |d alpha| under translation %.1e, rotation %.1e, scaling %.1e m" % (
        float(np.max(np.abs(al1 - al))), float(np.max(np.abs(al2 - al))), float(np.max(np.abs(al3 / k - al)))))


def test_long_ring_general_path_properties(gpu_engine):
    """N = 3000 (> 2048 waypoints: the interior point's general vector passes instead of the register-resident ones, the tridiagonal
    sweeps on workspace vectors instead of LDS): against CPU-B (independent assembly and solver), feasibility, start-index rotation.
    (The engine's self-reported KKT residual is only a sanity bound here: 1.2e-9 at this size since E is applied untruncated.)"""
    from oracle import banded_ref
    ref, nv, sc = synthetic.oval_batch(2, n=3000)
    probs = [dict(reftrack=ref[b], normvec=nv[b], scaling=sc[b], kappa_bound=0.12, w_veh=3.4) for b in range(2)]
    al, curv, st, info = gpu_engine.solve_batch(probs)
    assert np.all(st == 0)
    a_cpu, c_cpu, st_cpu, _, _ = banded_ref.solve_batch(ref, nv, sc, 0.12, 3.4)
    assert np.all(st_cpu == 0)
    for b in range(2):
        lo, hi = -(ref[b, :, 3] - 1.7), ref[b, :, 2] - 1.7
        assert np.all(al[b] >= lo - 1e-12) and np.all(al[b] <= hi + 1e-12)
        assert np.max(np.abs(al[b] - a_cpu[b])) < 1e-7 and abs(curv[b] - c_cpu[b]) < 1e-8
        assert info[b]["kkt_res"] < 5e-9 and 0 < info[b]["n_active_box"] < 3000
    r = 777
    pr = dict(reftrack=np.roll(ref[0], r, axis=0), normvec=np.roll(nv[0], r, axis=0), scaling=np.roll(sc[0], r),
              kappa_bound=0.12, w_veh=3.4)
    al2, _, st2, _ = gpu_engine.solve_batch([pr])
    assert st2[0] == 0 and np.max(np.abs(np.roll(al2[0], -r) - al[0])) < ALPHA_TOL


def test_very_long_rings_run_with_few_or_no_goldfarb_idnani_slots(monkeypatch):
    """the Goldfarb-Idnani slots -- a rare fallback -- were allocated eagerly and fatally: a ring of 20 000 waypoints needed
    6.4 GB for ONE slot and failed the launch with MCQ_E_DEVICE before any kernel ran, although round 4 had solved such rings through the
    long-ring route.  Round 6: full slots are capped by $MCQ_GI_BYTES and never fatal.  A ring of 20 000 waypoints with the default cap (two
    slots) and one of 40 000 with a cap no slot fits under (the handle then has NO Goldfarb-Idnani pool): both solve, within 1e-7 m of CPU-B
    (independent assembly and solver, O(n))."""
    from oracle import banded_ref
    for n, cap in ((20000, None), (40000, str(8 << 30))):
        if cap is None:
            monkeypatch.delenv("MCQ_GI_BYTES", raising=False)
        else:
            monkeypatch.setenv("MCQ_GI_BYTES", cap)           # 2 n^2 doubles = 25.6 GB per slot at n = 40 000: none fits
        eng = engine.Engine(0)
        try:
            xy = synthetic.oval_centreline(n, perimeter=3.0 * n, k1=7 * n // 2000, k2=23 * n // 2000, fine=20)
            nv, sc = synthetic.prepared_track(xy)
            ref = np.column_stack((xy, synthetic.widths(n, 77)))
            al, curv, st, info = eng.solve_batch([dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=0.12, w_veh=3.4)])
            assert st[0] == 0, (n, st[0], info[0])
            a_cpu, c_cpu, st_cpu, _, _ = banded_ref.solve_batch(ref[None], nv[None], sc[None], 0.12, 3.4)
            assert st_cpu[0] == 0 and np.max(np.abs(al[0] - a_cpu[0])) < 1e-7 and abs(curv[0] - c_cpu[0]) < 1e-8, (n, float(np.max(np.abs(al[0] - a_cpu[0]))))
            assert 0 < info[0]["n_active_box"] < n and info[0]["gi_iters"] == 0
            print("ring of %d waypoints: max |alpha - CPU-B| %.1e m, %d active rows, workspace %.2f GB" % (
                n, float(np.max(np.abs(al[0] - a_cpu[0]))), info[0]["n_active_box"], eng.workspace_bytes() / 1e9))
        finally:
            eng.close()


def test_oval_n1000_against_dense_gi_oracle(gpu_engine):
    """One N = 1000 oval against the LIVE dense oracle (dense inverse + dense Goldfarb-Idnani, about 10 s of CPU); the full
    size, N = 2000, is checked against the committed oracle output in test_oval_n2000_* below."""
    from oracle import qp_ref, tph_ref
    ref, nv, sc = synthetic.oval_batch(1, n=1000)
    path_cl = np.vstack((ref[0, :, :2], ref[0, :1, :2]))
    _, _, A, nv_d = tph_ref.calc_splines(path_cl)
    a_ref, err_ref = tph_ref.opt_min_curv(ref[0], nv_d, A, 0.12, 3.4)
    al, curv, st, _ = gpu_engine.solve_batch([dict(reftrack=ref[0], normvec=nv_d, scaling=sc[0], kappa_bound=0.12,
                                                   w_veh=3.4)])
    assert st[0] == 0
    assert np.max(np.abs(al[0] - a_ref)) < ALPHA_TOL
    assert abs(curv[0] - err_ref) < CURV_TOL
    assert qp_ref is not None


def test_fp32_boundary_full_size(gpu_engine):
    """BASELINE config 5's boundary at N = 2000 (perimeter 6 km, |x| up to 1.9 km): float rows / float alpha in HBM, fp64 arithmetic.
    STATED fp32 TOLERANCE OF CONFIG 5: |alpha(f32 rows) - alpha(f64 rows)| <= 1e-4 m, with the rows in the increment layout
    (MCQ_F32_INCREMENTS: float ring increments + fp64 origin; observed 5e-6 m, the size of rounding the WIDTH columns alone).
    Absolute float coordinates (MCQ_F32_ABSOLUTE, round 2's layout) lose the 3 m steps in the 1.2e-4 m ulp of a 1.9 km coordinate:
    2e-3 m -- measured here too, as the reason for the layout.  Both layouts are exact on the rows they rebuild: against the fp64
    entry fed those rows the only difference is the final rounding of alpha.  Device entry and host-buffer entry (mcq_solve_batch_f32)."""
    for pert in (False, True):
        ref, nv, sc = synthetic.oval_batch(4, n=2000, first=40, perturb_centreline=pert)
        full = [dict(reftrack=ref[b], normvec=nv[b], scaling=sc[b], kappa_bound=0.12, w_veh=3.4) for b in range(4)]
        a_full, _, st_f, _ = gpu_engine.solve_batch(full)
        assert np.all(st_f == 0)
        rows32, org = engine.rows_to_increments(ref)
        a_inc, curv_inc, st_i, info_i = gpu_engine.solve_batch_f32(rows32, org, 0.12, 3.4, layout=engine.F32_INCREMENTS)
        a_abs, _, st_a, _ = gpu_engine.solve_batch_f32(ref.astype(np.float32), None, 0.12, 3.4, layout=engine.F32_ABSOLUTE)
        assert a_inc.dtype == np.float32 and np.all(st_i == 0) and np.all(st_a == 0)
        dev_inc = max(float(np.max(np.abs(a_inc[b] - a_full[b]))) for b in range(4))
        dev_abs = max(float(np.max(np.abs(a_abs[b] - a_full[b]))) for b in range(4))
        print("fp32 boundary, N=2000, perturbed centrelines %s: max |alpha(f32 rows) - alpha(f64 rows)| = %.2e m (increments), %.2e m (absolute)"
              % (pert, dev_inc, dev_abs))
        assert dev_inc <= 1e-4                       # the stated tolerance
        assert dev_inc < 5e-5 and dev_abs > 5 * dev_inc
        # exact on the rebuilt rows
        r64 = engine.increments_to_rows(rows32, org)
        a64, curv64, st64, _ = gpu_engine.solve_batch([dict(reftrack=r64[b], normvec=None, scaling=None, kappa_bound=0.12, w_veh=3.4) for b in range(4)])
        assert np.all(st64 == 0)
        for b in range(4):
            assert np.max(np.abs(a_inc[b] - a64[b])) <= np.max(np.abs(a64[b])) * 2.0 ** -24 + 1e-9     # one rounding of alpha (+ summation order of the rebuild)
            assert abs(curv_inc[b] - curv64[b]) < 1e-10
            lo, hi = -(r64[b, :, 3] - 1.7), r64[b, :, 2] - 1.7
            assert np.all(a_inc[b] >= lo - 3e-7) and np.all(a_inc[b] <= hi + 3e-7)
            assert info_i[b].kkt_res < 1e-9
    # round 2's device entry (absolute rows) still answers, bitwise as the new entry in layout 0
    a_old, _, st_o, _ = gpu_engine.solve_uniform_f32(ref.astype(np.float32), None, None, 0.12, 3.4)
    assert np.all(st_o == 0) and np.array_equal(a_old, a_abs)


def test_config5_shard_shape_against_cpu_b_and_dense_goldens(gpu_engine):
    """BASELINE config 5 at one rank's shard SHAPE: 512 synthetic reference tracks, N = 2000, per-track
    centrelines (generator indices 0 .. 511 of config 5's generator), float increment rows in / float alpha out through the host-buffer
    entry mcq_solve_batch_f32.  Checked three ways: (i) every track against CPU-B (independent assembly and solver) run on the fp64 rows
    the device rebuilds from the float increments -- what is left is one rounding of alpha; (ii) the four tracks that have dense-oracle
    goldens (indices 5, 9: round 3; 13, 21: round 4) against those, within config 5's stated tolerance 1e-4 m (the goldens were solved on
    the fp64 rows); (iii) feasibility in the rebuilt box."""
    import os
    from oracle import banded_ref
    B, n = 512, 2000
    ref, nv, sc = synthetic.oval_batch(B, n=n, first=0, perturb_centreline=True)
    rows32, org = engine.rows_to_increments(ref)
    a32, curv, st, info = gpu_engine.solve_batch_f32(rows32, org, 0.12, 3.4, layout=engine.F32_INCREMENTS)
    assert a32.dtype == np.float32 and a32.shape == (B, n) and np.all(st == 0)
    r64 = engine.increments_to_rows(rows32, org)
    nv64 = np.empty((B, n, 2))
    sc64 = np.empty((B, n))
    for k in range(B):
        nv64[k], sc64[k] = synthetic.prepared_track(r64[k, :, :2])
    a_cpu, c_cpu, st_cpu, _, _ = banded_ref.solve_batch(r64, nv64, sc64, 0.12, 3.4)
    assert np.all(st_cpu == 0)
    err = np.max(np.abs(a32.astype(np.float64) - a_cpu), axis=1)
    assert err.max() < 1e-6, (int(np.argmax(err)), float(err.max()))          # one float rounding of |alpha| <= 4 m is 2.4e-7
    assert np.max(np.abs(curv - c_cpu)) < 1e-8
    lo, hi = -(r64[:, :, 3] - 1.7), r64[:, :, 2] - 1.7
    assert np.all(a32 >= lo - 3e-7) and np.all(a32 <= hi + 3e-7)
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    worst = 0.0
    for idx in (5, 9, 13, 21):
        g = np.load(os.path.join(gdir, "oval_n2000_c%d.npz" % idx))
        assert np.array_equal(g["reftrack"], ref[idx])
        worst = max(worst, float(np.max(np.abs(a32[idx].astype(np.float64) - g["alpha"]))))
    assert worst <= 1e-4 and worst < 5e-5, worst
    print("config 5 shard shape: 512 tracks, max |alpha_f32 - CPU-B(rebuilt rows)| = %.2e m, max |alpha_f32 - dense golden (fp64 rows)| = %.2e m"
          % (float(err.max()), worst))


def test_random_rings_against_dense_oracle(gpu_engine):
    """96 random star-shaped rings (n = 24 ... 160, widths between barely feasible and generous, so anything from a handful to
    most of the rows ends up on a bound) in one ragged launch, every one against the live dense oracle."""
    from oracle import tph_ref
    from test_emu_kernels import _small_track
    rng = np.random.default_rng(2024)
    probs, refs = [], []
    for k in range(96):
        n = int(rng.integers(24, 161))
        ref, nv, A, sc = _small_track(n, seed=7000 + k)
        w_veh = float(rng.choice([1.2, 2.0, 2.6]))
        ref[:, 2:] = 0.5 * w_veh + rng.uniform(0.05, 2.5) * rng.uniform(0.2, 1.0, size=(n, 2))
        probs.append(dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=1.0, w_veh=w_veh))
        refs.append(tph_ref.opt_min_curv(ref, nv, A, 1.0, w_veh))
    al, curv, st, info = gpu_engine.solve_batch(probs)
    worst = 0.0
    for k, (a_ref, err_ref) in enumerate(refs):
        assert st[k] == 0, (k, st[k])
        worst = max(worst, float(np.max(np.abs(al[k] - a_ref))))
        assert np.max(np.abs(al[k] - a_ref)) < ALPHA_TOL, (k, info[k])
        assert abs(curv[k] - err_ref) < CURV_TOL, k
    print("random rings: max |alpha - oracle| = %.2e m, active rows %d ... %d, pivoting rounds <= %d"
          % (worst, min(i["n_active_box"] for i in info), max(i["n_active_box"] for i in info), max(i["as_iters"] for i in info)))


def test_ragged_batch_and_small_rings(gpu_engine, golden):
    from oracle import tph_ref
    probs, refs = [], []
    for n in (7, 20, 64, 70, 129, 130):
        rng = np.random.default_rng(n)
        th = np.linspace(0.0, 2 * np.pi, n, endpoint=False)
        r = 40.0 + 6.0 * np.sin(3 * th + 1.0) + 3.0 * np.cos(5 * th + 2.0)
        xy = np.column_stack((r * np.cos(th), r * np.sin(th)))
        path_cl = np.vstack((xy, xy[0]))
        _, _, A, nv = tph_ref.calc_splines(path_cl)
        ref = np.column_stack((xy, 3.0 + rng.uniform(0.0, 1.5, size=(n, 2))))
        sc = tph.calc_splines.scalings_from_les_matrix(A)
        refs.append(tph_ref.opt_min_curv(ref, nv, A, 0.5, 2.0))
        probs.append(dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=0.5, w_veh=2.0))
    probs.append(_problem(golden["handling_track"]))
    al, curv, st, _ = gpu_engine.solve_batch(probs)
    assert np.all(st == 0)
    for k, (a_ref, err_ref) in enumerate(refs):
        assert np.max(np.abs(al[k] - a_ref)) < ALPHA_TOL
        assert abs(curv[k] - err_ref) < CURV_TOL
    assert np.max(np.abs(al[-1] - golden["handling_track"]["alpha"])) < ALPHA_TOL


def test_curvature_rows_active_and_infeasible_gpu(gpu_engine):
    from oracle import qp_ref, tph_ref
    n = 150
    rng = np.random.default_rng(11)
    th = np.linspace(0.0, 2 * np.pi, n, endpoint=False)
    r = 40.0 + 6.0 * np.sin(3 * th + 1.0) + 3.0 * np.cos(5 * th + 2.0)
    xy = np.column_stack((r * np.cos(th), r * np.sin(th)))
    path_cl = np.vstack((xy, xy[0]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    ref = np.column_stack((xy, 3.0 + rng.uniform(0.0, 1.5, size=(n, 2))))
    sc = tph.calc_splines.scalings_from_les_matrix(A)
    a_box, _, I = tph_ref.opt_min_curv(ref, nv, A, 10.0, 2.0, return_internals=True)
    kb = 0.9 * float(np.max(np.abs(I["k_ref"] + I["E"] @ a_box)))
    info = {}
    a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, kb, 2.0,
                                          solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
    n_act = int(np.sum(info["lagr"][2 * n:] > 0))
    assert n_act >= 1
    al, curv, st, inf = gpu_engine.solve_batch([dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=kb, w_veh=2.0),
                                                dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=1e-4, w_veh=2.0)])
    assert st[0] == 0 and st[1] == engine.STATUS_KAPPA_INFEASIBLE
    assert inf[0]["n_active_kappa"] == n_act
    assert np.max(np.abs(al[0] - a_ref)) < ALPHA_TOL
    assert abs(curv[0] - err_ref) < CURV_TOL
    with pytest.raises(ValueError, match="inconsistent"):
        tph.opt_min_curv.opt_min_curv(ref, nv, A, 1e-4, 2.0)


def test_vehicle_width_sweep_mixed_tracks(gpu_engine, golden):
    """BASELINE config 4 in miniature: one ragged launch over (track x vehicle-width) variants of the reference tracks
    (re-sampled to N ~ 250 so that the dense oracle finishes in seconds), every variant against the live dense oracle."""
    from oracle import tph_ref
    probs, refs = [], []
    for name in ("berlin_2018", "modena_2019", "rounded_rectangle"):
        full = golden[name]["reftrack"]
        n_sub = min(250, full.shape[0])
        idx = np.round(np.linspace(0, full.shape[0], n_sub, endpoint=False)).astype(int)
        ref = full[idx]
        path_cl = np.vstack((ref[:, :2], ref[0, :2]))
        _, _, A, nv = tph_ref.calc_splines(path_cl)
        sc = tph.calc_splines.scalings_from_les_matrix(A)
        for w_veh in (2.0, 2.8, 3.4):
            if np.any(ref[:, 2] + ref[:, 3] < w_veh):
                continue                                  # infeasible variant: covered by the status-code test
            probs.append(dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=0.12, w_veh=w_veh))
            refs.append(tph_ref.opt_min_curv(ref, nv, A, 0.12, w_veh))
    assert len(probs) >= 6
    al, curv, st, info = gpu_engine.solve_batch(probs)
    for k, (a_ref, err_ref) in enumerate(refs):
        assert st[k] == 0, (k, st[k])
        assert np.max(np.abs(al[k] - a_ref)) < ALPHA_TOL, k
        assert abs(curv[k] - err_ref) < CURV_TOL, k


def test_device_ragged_entry_with_per_problem_vehicle_parameters(gpu_engine, golden):
    """mcq_solve_device_ragged_params (resident tracks, per-problem w_veh / kappa_bound as device arrays) against the
    host-buffer entry, which carries the same parameters in mcq_problem: bitwise the same alpha."""
    names = ("handling_track", "rounded_rectangle", "modena_2019")
    probs = []
    for k, name in enumerate(names):
        for w_veh, kb in ((2.0, 0.12), (3.0, 0.2)):
            g = golden[name]
            probs.append(dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=kb, w_veh=w_veh))
    al_h, curv_h, st_h, _ = gpu_engine.solve_batch(probs)
    bsz = len(probs)
    ns = np.array([p["reftrack"].shape[0] for p in probs], dtype=np.int32)
    nmax = int(ns.max())
    ref = np.zeros((bsz, nmax, 4)); nv = np.zeros((bsz, nmax, 2)); sc = np.ones((bsz, nmax))
    for k, p in enumerate(probs):
        ref[k, :ns[k]] = p["reftrack"]; nv[k, :ns[k]] = p["normvec"]; sc[k, :ns[k]] = p["scaling"]
    eng = gpu_engine
    ptrs = []

    def up(a):
        ptrs.append(eng.alloc(a.nbytes))
        eng.upload(ptrs[-1], a)
        return ptrs[-1]
    try:
        d_ref, d_nv, d_sc, d_n = up(ref), up(nv), up(sc), up(ns)
        d_kb = up(np.array([p["kappa_bound"] for p in probs]))
        d_wv = up(np.array([p["w_veh"] for p in probs]))
        d_al, d_cu, d_st = up(np.zeros((bsz, nmax))), up(np.zeros(bsz)), up(np.zeros(bsz, dtype=np.int32))
        eng.solve_device_ragged_params(bsz, nmax, d_n, d_ref, d_nv, d_sc, 0.0, 0.0, d_kb, d_wv, d_al, d_cu, d_st)
        eng.sync()
        al_d = eng.download(d_al, (bsz, nmax), np.float64)
        st_d = eng.download(d_st, (bsz,), np.int32)
        cu_d = eng.download(d_cu, (bsz,), np.float64)
    finally:
        for p in ptrs:
            eng.free(p)
    assert np.array_equal(st_d, st_h) and np.all(st_h == 0)
    for k in range(bsz):
        assert np.array_equal(al_d[k, :ns[k]], al_h[k]), k
        assert cu_d[k] == curv_h[k]


def test_iqp_device_resident_matches_golden(gpu_engine, golden):
    """Row f-1: a batch of IQP runs with the tracks resident in HBM between the passes (mcq_solve_device_ragged +
    mcq_relinearise_device) against the golden IQP end states of the oracle's host chain; ragged N, tracks finishing in
    different rounds."""
    names = ("rounded_rectangle", "handling_track", "rounded_rectangle")
    tracks = []
    for name in names:
        g = golden[name]
        tracks.append(dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"]))
    stats = {}
    out = tph.iqp_handler.iqp_handler_batch(tracks, kappa_bound=0.12, w_veh=3.4, stepsize_interp=3.0, iters_min=3,
                                            curv_error_allowed=0.01, engine=gpu_engine, stats=stats, device_resident=True)
    assert stats["device_resident"] and stats["rounds"] >= 3
    for name, (a, ref_out, nv_out) in zip(names, out):
        g = golden[name]
        assert a.shape == g["iqp_alpha"].shape
        assert np.max(np.abs(a - g["iqp_alpha"])) < ALPHA_TOL
        assert np.max(np.abs(ref_out - g["iqp_reftrack"])) < 1e-6
        assert np.max(np.abs(nv_out - g["iqp_normvec"])) < 1e-8


def test_iqp_device_resident_driver_bulk_round(gpu_engine):
    """The device-resident IQP driver (QP pass + glue kernel per round, tracks resident between passes) against the host-glue
    driver on a batch large enough that the tracks finishing in one round come back with the three bulk copies."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler as iq
    from test_emu_kernels import _small_track
    tracks = []
    for k in range(12):
        ref, nv, _, sc = _small_track(18 + k, seed=500 + k)
        tracks.append(dict(reftrack=ref, normvectors=nv, scaling=sc))
    step = 2.0 * np.pi * 40.0 / 22.0
    st_d, st_h = {}, {}
    out_d = iq.iqp_handler_batch([dict(t, reftrack=t["reftrack"].copy()) for t in tracks], 0.5, 2.0, step, 3, 0.01, engine=gpu_engine,
                                 stats=st_d, device_resident=True)
    out_h = iq.iqp_handler_batch([dict(t, reftrack=t["reftrack"].copy()) for t in tracks], 0.5, 2.0, step, 3, 0.01, engine=gpu_engine,
                                 stats=st_h, device_resident=False)
    assert st_d["rounds"] == st_h["rounds"] and st_d["qp_solves"] == st_h["qp_solves"]
    for (a_d, r_d, n_d), (a_h, r_h, n_h) in zip(out_d, out_h):
        assert a_d.shape == a_h.shape and r_d.shape == r_h.shape
        assert np.max(np.abs(a_d - a_h)) < 1e-8
        assert np.max(np.abs(r_d - r_h)) < 1e-8 and np.max(np.abs(n_d - n_h)) < 1e-8


def test_iqp_warm_start_equals_cold_start_full_size(gpu_engine):
    """32 synthetic N = 2000 ovals through the device-resident IQP driver twice: passes 2+ warm-started from the previous
    pass's working set (exchange only) and cold (interior point every pass).  Both end at exact KKT vertices of the same QPs:
    the end states agree to the conditioning of the reduced systems, far inside the 1e-6 m tolerance."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler as iq
    ref, nv, sc = synthetic.oval_batch(32, n=2000, first=600)
    tracks = [dict(reftrack=ref[k], normvectors=nv[k], scaling=sc[k]) for k in range(32)]
    outs = []
    for warm in (True, False):
        st = {}
        outs.append(iq.iqp_handler_batch([dict(t, reftrack=t["reftrack"].copy()) for t in tracks], 0.12, 3.4, 3.0, 3, 0.01,
                                         engine=gpu_engine, stats=st, device_resident=True, warm_start=warm))
        assert st["rounds"] == 3
    worst = 0.0
    for (a_w, r_w, n_w), (a_c, r_c, n_c) in zip(*outs):
        assert a_w.shape == a_c.shape
        worst = max(worst, float(np.max(np.abs(a_w - a_c))))
        assert np.max(np.abs(a_w - a_c)) < ALPHA_TOL and np.max(np.abs(r_w - r_c)) < ALPHA_TOL
    print("IQP warm vs cold start, 32 x N=2000: max |alpha| difference %.2e m" % worst)


def test_degenerate_iqp_pass_two_attempt_driver(gpu_engine):
    """The QP of the third IQP pass on a synthetic N = 2000 oval (tests/golden/iqp_pass3_oval3.npz, made by
    scripts/make_degenerate_fixture.py with the dense oracle): 42 touched bounds with multipliers down to 1e-7 of the
    gradient scale.  The interior point at mu = 1e-10 is 5 mm from the optimum here and block pivoting from its guess does
    not settle; the driver's second attempt (interior point resumed to mu = 1e-13) must return the oracle's vertex."""
    from conftest import load_golden
    g = load_golden("iqp_pass3_oval3")
    al, curv, st, info = gpu_engine.solve_batch([dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=None,
                                                      kappa_bound=float(g["kappa_bound"]), w_veh=float(g["w_veh"]))])
    assert st[0] == 0
    assert np.max(np.abs(al[0] - g["alpha"])) < ALPHA_TOL
    assert abs(curv[0] - float(g["curv_error_max"])) < CURV_TOL
    assert info[0]["kkt_res"] < 1e-9
    assert info[0]["as_iters"] <= 12


def test_block_pivoting_pins_one_row_per_neighbourhood(gpu_engine):
    """Third IQP pass of synthetic oval 629 (tests/golden/iqp_pass3_oval629.npz): with every row of a stretch that leaves the
    box pinned at once, block pivoting needed 49 rounds on this instance even from the mu = 1e-13 guess, which is off by ONE
    row (one slow problem stretches the whole launch: 97 instead of 66 ms for the 1024 third passes).  Pinning only the
    furthest-out row per neighbourhood settles it -- and the oval-3 instance -- in a handful of rounds."""
    from conftest import load_golden
    for name in ("iqp_pass3_oval629", "iqp_pass3_oval3"):
        g = load_golden(name)
        al, curv, st, info = gpu_engine.solve_batch([dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=None,
                                                          kappa_bound=float(g["kappa_bound"]), w_veh=float(g["w_veh"]))])
        assert st[0] == 0
        assert np.max(np.abs(al[0] - g["alpha"])) < ALPHA_TOL
        assert abs(curv[0] - float(g["curv_error_max"])) < CURV_TOL
        assert info[0]["as_iters"] <= 8, (name, info[0]["as_iters"])


def test_prep_on_device_and_solve_without_normals(gpu_engine, golden):
    """Row f-2: normals / scalings derived on the device against tph.calc_splines (as prep_track calls it), and the
    reference tracks solved from [x, y, w_right, w_left] rows alone against the golden alpha."""
    names = list(golden)
    nvs, scs = gpu_engine.prep_batch([golden[k]["reftrack"] for k in names])
    for k, name in enumerate(names):
        g = golden[name]
        assert np.max(np.abs(nvs[k] - g["normvec"])) < 1e-10, name
        assert np.max(np.abs(scs[k] - g["scaling"])) < 1e-12, name
    al, curv, st, _ = gpu_engine.solve_batch([dict(reftrack=golden[k]["reftrack"], normvec=None, scaling=None,
                                                   kappa_bound=float(golden[k]["kappa_bound"]), w_veh=float(golden[k]["w_veh"]))
                                              for k in names])
    for k, name in enumerate(names):
        assert st[k] == 0
        assert np.max(np.abs(al[k] - golden[name]["alpha"])) < ALPHA_TOL, name
        assert abs(curv[k] - float(golden[name]["curv_error_max"])) < CURV_TOL, name


def test_velocity_profile_filter_window_and_friction_map(gpu_engine, golden):
    """mcq_vel_profile_device_opts on the GPU: tph.calc_vel_profile's filt_window [REF params/racecar.ini:54-57,
    main_globaltraj.py:407] and per-waypoint friction coefficients against oracle/vel_ref.py and the host shim (the helper of the
    emulator suite, here through the real library)."""
    from test_emu_kernels import _check_vel_profiles, _raceline_kappa_el, _vehicle_variants, _vehicle_variants_speed_dependent
    for name in ("rounded_rectangle", "modena_2019"):
        kappa, el = _raceline_kappa_el(golden[name])
        n = kappa.size
        mu = 0.9 + 0.2 * np.cos(2.0 * np.pi * np.arange(n) / n * 3.0)
        _check_vel_profiles(gpu_engine, kappa, el, _vehicle_variants(), 1.0, filt_window=5)
        _check_vel_profiles(gpu_engine, kappa, el, _vehicle_variants_speed_dependent(), 2.0, mu=mu)
        _check_vel_profiles(gpu_engine, kappa, el, _vehicle_variants_speed_dependent(), 1.0, mu=mu, filt_window=7)


def test_velocity_profile_lap_time_sweep(gpu_engine, golden):
    """Row f-3: a (gg-scale x top-speed) grid of vehicle variants over the racelines of two reference tracks in ONE launch
    (what the reference's lap-time matrix loops over [REF main_globaltraj.py:442-496]) against the ORACLE's restatement of
    tph.calc_vel_profile -> calc_ax_profile -> calc_t_profile (oracle/vel_ref.py: acceleration-phase gating, backward
    look-ahead, all ggv rows), variant by variant; the constant ggv of the reference scaled as the sweep scales it, plus
    speed-dependent diagrams with top speeds between the grid points."""
    from oracle import vel_ref
    from test_emu_kernels import _vehicle_variants_speed_dependent
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import create_raceline as cr, calc_head_curv_an as ch
    tracks = []
    for name in ("handling_track", "modena_2019"):
        g = golden[name]
        out = cr.create_raceline(refline=g["reftrack"][:, :2], normvectors=g["normvec"], alpha=g["alpha"], stepsize_interp=3.0)
        _, kappa = ch.calc_head_curv_an(coeffs_x=out[2], coeffs_y=out[3], ind_spls=out[4], t_spls=out[5])
        tracks.append((kappa, out[8]))
    nmax = max(k.size for k, _ in tracks)
    v = np.arange(0.0, 72.1, 4.0)
    ggv0 = np.column_stack((v, np.full(v.size, 12.0), np.full(v.size, 12.0)))
    axm = np.column_stack((v, np.interp(v, [0.0, 20.0, 72.0], [5.3, 5.3, 1.2])))
    for kappa, el in tracks:                      # one launch per track length (uniform n per launch)
        ggvs, vmaxs = [], []
        for scale in (0.3, 0.65, 1.0):
            for top in (100.0 / 3.6, 150.0 / 3.6, 70.0):
                gg = ggv0.copy()
                gg[:, 1:] *= scale
                ggvs.append(gg)
                vmaxs.append(top)
        for gg, _, _, _, top in _vehicle_variants_speed_dependent():
            ggvs.append(gg)
            vmaxs.append(top)
        bsz = len(ggvs)
        for e in (1.0, 1.7):
            vx_d, lt_d = gpu_engine.vel_profile_batch(kappa[None, :], el[None, :], np.stack(ggvs), np.stack([axm] * bsz), 0.75,
                                                      1200.0, vmaxs, dyn_model_exp=e, track_of=np.zeros(bsz, dtype=np.int32))
            for k in range(bsz):
                vx_o = vel_ref.calc_vel_profile(ax_max_machines=axm, kappa=kappa, el_lengths=el, closed=True, drag_coeff=0.75,
                                                m_veh=1200.0, ggv=ggvs[k], v_max=vmaxs[k], dyn_model_exp=e)
                ax_o = vel_ref.calc_ax_profile(np.append(vx_o, vx_o[0]), el)
                t_o = vel_ref.calc_t_profile(vx_o, el, ax_profile=ax_o)
                assert np.max(np.abs(vx_d[k] - vx_o)) < 1e-9, (k, e)
                # lap time: the device sums 2 l / (v_a + v_b) per element -- algebraically calc_t_profile's expression, which
                # cancels catastrophically on speed-limited stretches (a -> 0); hence exact against the stable form, loose
                # against upstream's
                assert abs(lt_d[k] - vel_ref.lap_time_stable(vx_o, el)) < 1e-9, (k, e)
                assert abs(lt_d[k] - t_o[-1]) < 0.5, (k, e)
    assert nmax > 0


def test_raceline_kernel_and_ragged_lap_time_matrix(gpu_engine, golden):
    """The chain after the QP [REF main_globaltraj.py:371-422] on the device over the four reference tracks at once (BASELINE
    config 4's shape): alpha from the engine -> mcq_raceline_device (create_raceline + calc_head_curv_an) -> ragged velocity
    profiles of a (gg-scale x top-speed) grid per raceline; every raceline against the host shims (themselves checked against
    oracle/tph_ref.py in tests/test_host.py), every velocity profile / lap time against oracle/vel_ref.py."""
    from oracle import vel_ref
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import create_raceline as cr, calc_head_curv_an as ch
    names = ("berlin_2018", "modena_2019", "handling_track", "rounded_rectangle")
    probs = [_problem(golden[k]) for k in names]
    al, _, st, _ = gpu_engine.solve_batch(probs)
    assert np.all(st == 0)
    out = gpu_engine.raceline_batch([p["reftrack"] for p in probs], [p["normvec"] for p in probs], al, 2.0)
    assert np.all(out["status"] == 0)
    host = []
    for k, p in enumerate(probs):
        rl, _, cx, cy, inds, tv, _, _, el = cr.create_raceline(p["reftrack"][:, :2], p["normvec"], al[k], 2.0)
        psi, kap = ch.calc_head_curv_an(cx, cy, inds, tv)
        m = int(out["m"][k])
        assert m == rl.shape[0]
        dpsi = np.abs(out["psi"][k, :m] - psi)
        assert np.max(np.abs(out["xy"][k, :m] - rl)) < 1e-9
        assert np.max(np.minimum(dpsi, 2 * np.pi - dpsi)) < 1e-10
        assert np.max(np.abs(out["kappa"][k, :m] - kap)) < 1e-11
        assert np.max(np.abs(out["el_lengths"][k, :m] - el)) < 1e-9
        host.append((kap, el))
    v = np.arange(0.0, 72.1, 4.0)
    ggv0 = np.column_stack((v, np.full(v.size, 12.0), np.full(v.size, 12.0)))
    axm = np.column_stack((v, np.interp(v, [0.0, 20.0, 72.0], [5.3, 5.3, 1.2])))
    ggvs, tops, track_of = [], [], []
    for t in range(4):
        for scale in (0.3, 1.0):
            for top in (100.0 / 3.6, 70.0):
                gg = ggv0.copy()
                gg[:, 1:] *= scale
                ggvs.append(gg)
                tops.append(top)
                track_of.append(t)
    bsz = len(ggvs)
    vx_d, lt_d = gpu_engine.vel_profile_batch(out["kappa"], out["el_lengths"], np.stack(ggvs), np.stack([axm] * bsz), 0.75, 1200.0,
                                              tops, dyn_model_exp=1.0, track_of=np.array(track_of, dtype=np.int32),
                                              n_of_track=out["m"])
    for k in range(bsz):
        kap, el = host[track_of[k]]
        vx_o = vel_ref.calc_vel_profile(ax_max_machines=axm, kappa=kap, el_lengths=el, closed=True, drag_coeff=0.75, m_veh=1200.0,
                                        ggv=ggvs[k], v_max=tops[k], dyn_model_exp=1.0)
        assert np.max(np.abs(vx_d[k, :kap.size] - vx_o)) < 1e-8, k
        assert abs(lt_d[k] - vel_ref.lap_time_stable(vx_o, el)) < 1e-8, k


def test_normals_crossing_on_device(gpu_engine, golden):
    """Row f-2's second half: tph.check_normals_crossing [REF helper_funcs_glob/src/prep_track.py:57-59] for a ragged batch,
    against the oracle's restatement (oracle/vel_ref.py)."""
    from oracle import vel_ref as cn
    from test_emu_kernels import _crossing_cases
    cases = _crossing_cases(golden) + [(golden[k]["reftrack"], golden[k]["normvec"]) for k in ("berlin_2018", "handling_track")]
    got = gpu_engine.normals_crossing_batch([c[0] for c in cases], [c[1] for c in cases], horizon=10)
    for k, (t, nv) in enumerate(cases):
        if t.shape[0] <= 10:
            assert got[k] == -1
        else:
            assert got[k] == int(cn.check_normals_crossing(t, nv, 10)), k


def test_pinned_variables_and_bad_input(gpu_engine, golden):
    """Edge cases of the boundary: waypoints whose box is a single point (w_right + w_left == w_veh: the interior point
    carries them as pinned rows, the masked factorisation path) against the dense oracle, and non-finite input flagged
    per problem (status 4) without disturbing its batch neighbours."""
    from oracle import tph_ref
    g = golden["rounded_rectangle"]
    ref = g["reftrack"].copy()
    for i, shift in ((5, 0.2), (40, -0.35), (41, 0.1)):
        ref[i, 2] = 1.7 + shift          # hi =  shift
        ref[i, 3] = 1.7 - shift          # lo =  shift
    path_cl = np.vstack((ref[:, :2], ref[0, :2]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    # the oracle with the three zero-width waypoints stated as EQUALITIES (meq = 3) -- on tph's two-inequality form quadprog's own rule set ends
    # in a spurious "constraints are inconsistent" on this input (a rounding residue of -4.5e-15 on the partner row of an active one:
    # oracle/qp_ref.solve_qp_gi_zero_width_as_equalities has the mechanism); the engine returns the vertex
    from oracle import qp_ref
    a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, 0.12, 3.4, solver=qp_ref.solve_qp_gi_zero_width_as_equalities)
    with pytest.raises(ValueError, match="inconsistent"):
        tph_ref.opt_min_curv(ref, nv, A, 0.12, 3.4)
    bad = g["reftrack"].copy()
    bad[7, 0] = np.nan
    sc = tph.calc_splines.scalings_from_les_matrix(A)
    al, curv, st, _ = gpu_engine.solve_batch([dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=0.12, w_veh=3.4),
                                              dict(reftrack=bad, normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12, w_veh=3.4),
                                              dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12, w_veh=3.4)])
    assert list(st) == [0, engine.STATUS_BAD_INPUT, 0]
    assert np.max(np.abs(al[0] - a_ref)) < ALPHA_TOL
    assert abs(al[0][5] - 0.2) < 1e-12 and abs(al[0][40] + 0.35) < 1e-12
    assert abs(curv[0] - err_ref) < CURV_TOL
    assert np.max(np.abs(al[2] - g["alpha"])) < ALPHA_TOL


def _sp_kkt(ref, nv, w_veh, alpha):
    """Projected-gradient certificate of the shortest-path QP from its tridiagonal form (numpy, O(n))."""
    p = ref[:, :2]
    hd = 4.0 * np.sum(nv * nv, axis=1)
    ho = -2.0 * np.sum(nv * np.roll(nv, -1, axis=0), axis=1)
    g = hd * alpha + ho * np.roll(alpha, -1) + np.roll(ho, 1) * np.roll(alpha, 1) \
        + 2.0 * np.sum(nv * (2 * p - np.roll(p, 1, axis=0) - np.roll(p, -1, axis=0)), axis=1)
    lo, hi = -np.maximum(ref[:, 3] - w_veh / 2, 0.001), np.maximum(ref[:, 2] - w_veh / 2, 0.001)
    at_lo, at_hi = alpha <= lo + 1e-12, alpha >= hi - 1e-12
    free = ~(at_lo | at_hi)
    viol = max(float(np.max(np.abs(g[free]), initial=0.0)), float(np.max(-g[at_lo], initial=0.0)),
               float(np.max(g[at_hi], initial=0.0)))
    feas = max(float(np.max(lo - alpha)), float(np.max(alpha - hi)))
    return viol, feas, int(np.count_nonzero(~free))


def test_shortest_path_drop_in_matches_golden(golden):
    """Row f-4: tph.opt_shortest_path drop-in [REF main_globaltraj.py:286-290] on the four reference tracks."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "shortest_path.npz"))
    for name in ("rounded_rectangle", "handling_track", "modena_2019", "berlin_2018"):
        g = golden[name]
        a = tph.opt_shortest_path.opt_shortest_path(reftrack=g["reftrack"], normvectors=g["normvec"],
                                                    w_veh=float(z["w_veh"]), print_debug=False)
        assert a.shape == (g["reftrack"].shape[0],)
        assert np.max(np.abs(a - z[name + "_alpha"])) < ALPHA_TOL
    with pytest.raises(RuntimeError, match="same as normvectors"):
        tph.opt_shortest_path.opt_shortest_path(g["reftrack"], g["normvec"][:-1], 3.4)


def test_shortest_path_full_size_properties_and_oracle(gpu_engine):
    """N = 2000 ovals: KKT certificate (numpy, tridiagonal form), feasibility, shorter polygon, batch-order independence;
    one N = 1000 problem against the dense Goldfarb-Idnani oracle."""
    from oracle import tph_ref
    ref, nv, _ = synthetic.oval_batch(8, n=2000, perturb_centreline=True)
    probs = [dict(reftrack=ref[b], normvec=nv[b], scaling=None, kappa_bound=1.0, w_veh=3.4) for b in range(8)]
    al, curv, st, info = gpu_engine.solve_batch(probs, objective=engine.OBJ_SHORTEST_PATH)
    assert np.all(st == 0) and np.all(curv == 0.0)
    for b in range(8):
        viol, feas, nact = _sp_kkt(ref[b], nv[b], 3.4, al[b])
        assert viol < 1e-8 and feas < 1e-12
        assert nact == info[b]["n_active_box"] and 0 < nact < 2000
        assert tph_ref.path_length_sq(ref[b], nv[b], al[b]) < tph_ref.path_length_sq(ref[b], nv[b], np.zeros(2000))
    al2, _, _, _ = gpu_engine.solve_batch(probs[::-1], objective=engine.OBJ_SHORTEST_PATH)
    for b in range(8):
        assert np.array_equal(al[b], al2[7 - b])
    ref1, nv1, _ = synthetic.oval_batch(1, n=1000, first=3, perturb_centreline=True)
    a_ref = tph_ref.opt_shortest_path(ref1[0], nv1[0], 3.4)
    a, _, st1, _ = gpu_engine.solve_batch([dict(reftrack=ref1[0], normvec=nv1[0], scaling=None, kappa_bound=1.0,
                                                w_veh=3.4)], objective=engine.OBJ_SHORTEST_PATH)
    assert st1[0] == 0
    assert np.max(np.abs(a[0] - a_ref)) < ALPHA_TOL
    # ragged batch around the switches of the scalar tridiagonal route: one row per thread (n <= 256), partial last blocks, the
    # workspace-vector route (n > 2048)
    sizes = [100, 256, 257, 777, 2048, 2049, 2600, 4100]
    probs = []
    for k, n in enumerate(sizes):
        r, v, _ = synthetic.oval_batch(1, n=n, first=300 + k, perturb_centreline=True)
        probs.append(dict(reftrack=r[0], normvec=v[0], scaling=None, kappa_bound=1.0, w_veh=2.8))
    al3, _, st3, info3 = gpu_engine.solve_batch(probs, objective=engine.OBJ_SHORTEST_PATH)
    assert np.all(st3 == 0)
    for k, pr in enumerate(probs):
        viol, feas, nact = _sp_kkt(pr["reftrack"], pr["normvec"], 2.8, al3[k])
        assert viol < 1e-8 and feas < 1e-12 and nact == info3[k]["n_active_box"]


def test_mintime_reopt_corridor_config(gpu_engine, golden):
    """The second consumer of opt_min_curv in the reference: the re-optimisation after the minimum-time run
    [REF main_globaltraj.py:337-350] with `w_tr_reopt = 2.0` -> widths 1.0 / 1.0 and `w_veh_reopt = 1.6`
    [REF params/racecar.ini:110-111]: a uniform +-0.2 m corridor around the line.  Same entry point, a very different active
    fraction (14-28 % of the box rows, and on Berlin two curvature rows at the optimum): every golden track against the live
    dense oracle (dense inverse + dense Goldfarb-Idnani with all 4N rows)."""
    from oracle import tph_ref
    probs, refs = [], []
    for name, g in golden.items():
        ref = g["reftrack"].copy()
        ref[:, 2:] = 0.5 * 2.0
        A = tph.calc_splines.build_les_matrix(ref.shape[0], g["scaling"])
        refs.append((name, ref, tph_ref.opt_min_curv(ref, g["normvec"], A, 0.12, 1.6)))
        probs.append(dict(reftrack=ref, normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12, w_veh=1.6))
    al, curv, st, info = gpu_engine.solve_batch(probs)
    n_kappa = 0
    for k, (name, ref, (a_ref, err_ref)) in enumerate(refs):
        assert st[k] == 0, (name, st[k])
        assert np.max(np.abs(al[k] - a_ref)) < ALPHA_TOL, (name, float(np.max(np.abs(al[k] - a_ref))))
        assert abs(curv[k] - err_ref) < CURV_TOL, name
        assert np.all(np.abs(al[k]) <= 0.2 + 1e-12)
        assert info[k]["n_active_box"] == int(np.sum(np.abs(np.abs(a_ref) - 0.2) < 1e-9)), name
        n_kappa += info[k]["n_active_kappa"]
    assert n_kappa >= 2          # Berlin: the corridor leaves two curvature rows at the bound
    # and through the drop-in function, as main_globaltraj.py calls it
    name, ref, (a_ref, _) = refs[-1]
    g = golden[name]
    A = tph.calc_splines.build_les_matrix(ref.shape[0], g["scaling"])
    a = tph.opt_min_curv.opt_min_curv(reftrack=ref, normvectors=g["normvec"], A=A, kappa_bound=0.12, w_veh=1.6,
                                      print_debug=False, plot_debug=False)[0]
    assert np.max(np.abs(a - a_ref)) < ALPHA_TOL


def _golden_n2000():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "oval_n2000.npz"))
    g = {k: z[k] for k in z.files}
    # the fixture's inputs ARE the BASELINE config 3 generator's track 0 (what bench.py solves 1024 width variants of)
    ref, nv, sc = synthetic.oval_batch(1, n=2000)
    assert np.array_equal(ref[0], g["reftrack"]) and np.array_equal(nv[0], g["normvec"]) and np.array_equal(sc[0], g["scaling"])
    return g


def test_oval_n2000_first_pass_against_golden(gpu_engine):
    """BASELINE config 3 at full size: alpha and curv_error_max of one opt_min_curv pass at N = 2000 against the committed
    output of the dense-faithful oracle (8000 x 8000 dense inverse, dense Goldfarb-Idnani with all 8000 rows;
    scripts/make_golden_n2000.py, pinned there by the trust-region-reflective second route to 3.5e-10 m and by a KKT
    certificate).  Host-buffer entry, device entry fed rows only (normals derived on the device), and the drop-in function."""
    g = _golden_n2000()
    p = dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12, w_veh=3.4)
    for prob in (p, dict(p, normvec=None, scaling=None)):
        al, curv, st, info = gpu_engine.solve_batch([prob])
        assert st[0] == 0
        assert np.max(np.abs(al[0] - g["alpha"])) < ALPHA_TOL, float(np.max(np.abs(al[0] - g["alpha"])))
        assert abs(curv[0] - float(g["curv_error_max"])) < CURV_TOL
        assert abs(info[0]["kappa_max"] - float(g["kappa_max"])) < 1e-9
    A = tph.calc_splines.build_les_matrix(2000, g["scaling"])
    a, err = tph.opt_min_curv.opt_min_curv(g["reftrack"], g["normvec"], A, 0.12, 3.4)
    assert np.max(np.abs(a - g["alpha"])) < ALPHA_TOL and abs(err - float(g["curv_error_max"])) < CURV_TOL


def test_oval_n2000_more_width_seeds_and_config5_tracks_against_golden(gpu_engine):
    """Four more full-size problems through the dense oracle (scripts/make_golden_r3.py; each pinned by the TRF second route to
    < 1e-9 m and a KKT certificate): width seeds 1 and 2 of the bench workload (BASELINE config 3) and generator indices 5 and 9
    of config 5's generator (perturb_centreline = True: per-track centrelines).  The fixtures' inputs must BE the generator's
    output; alpha / curv_error_max against the oracle's, from rows + normals + scalings and from rows alone."""
    from conftest import load_golden
    probs, want = [], []
    for name, idx, pert in (("oval_n2000_w1", 1, False), ("oval_n2000_w2", 2, False), ("oval_n2000_c5", 5, True), ("oval_n2000_c9", 9, True)):
        g = load_golden(name)
        ref, nv, sc = synthetic.oval_batch(1, n=2000, first=idx, perturb_centreline=pert)
        assert np.array_equal(ref[0], g["reftrack"]) and np.array_equal(nv[0], g["normvec"]) and np.array_equal(sc[0], g["scaling"]), name
        probs.append(dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12, w_veh=3.4))
        probs.append(dict(reftrack=g["reftrack"], normvec=None, scaling=None, kappa_bound=0.12, w_veh=3.4))
        want += [(name, g), (name, g)]
    al, curv, st, info = gpu_engine.solve_batch(probs[0::2])
    al2, curv2, st2, info2 = gpu_engine.solve_batch(probs[1::2])
    worst = 0.0
    for k, (name, g) in enumerate(want[0::2]):
        for a, c, s in ((al[k], curv[k], st[k]), (al2[k], curv2[k], st2[k])):
            assert s == 0, name
            worst = max(worst, float(np.max(np.abs(a - g["alpha"]))))
            assert np.max(np.abs(a - g["alpha"])) < ALPHA_TOL, (name, float(np.max(np.abs(a - g["alpha"]))))
            assert abs(c - float(g["curv_error_max"])) < CURV_TOL, name
    print("N=2000 goldens (2 width seeds, 2 config-5 tracks): max |alpha - dense oracle| = %.2e m" % worst)


def test_round4_dense_goldens_at_bench_size(gpu_engine):
    """Eight more N = 2000 problems through the dense oracle (scripts/make_golden_r4.py, tests/golden/SUMMARY_r4.json):
    three width seeds of the bench workload, two config-5 tracks, the QPs of an IQP second pass (oval 5) and third pass (oval 9) -- rings
    of 2003 / 2002 waypoints with unit scalings and dozens of barely active bounds --, and a ring whose CURVATURE bound is active at the
    optimum at this size (box rows and curvature rows in one working set, 0.93 of the box optimum's curvature maximum)."""
    from conftest import load_golden
    worst = {}
    for name in ("oval_n2000_w3", "oval_n2000_w7", "oval_n2000_w11", "oval_n2000_c13", "oval_n2000_c21", "iqp_pass2_oval5", "iqp_pass3_oval9",
                 "oval_n2000_kappa"):
        g = load_golden(name)
        sc = g["scaling"] if "scaling" in g else None
        kb = float(g["kappa_bound"])
        if name.startswith("oval_n2000_") and name != "oval_n2000_kappa":
            ref, nv, sc_g = synthetic.oval_batch(1, n=2000, first=int(g["generator_index"]), perturb_centreline=bool(g["perturb_centreline"]))
            assert np.array_equal(ref[0], g["reftrack"]) and np.array_equal(nv[0], g["normvec"]) and np.array_equal(sc_g[0], sc), name
        al, curv, st, info = gpu_engine.solve_batch([dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=sc, kappa_bound=kb, w_veh=3.4)])
        assert st[0] == 0, (name, st[0])
        worst[name] = float(np.max(np.abs(al[0] - g["alpha"])))
        assert worst[name] < ALPHA_TOL, (name, worst[name])
        assert abs(curv[0] - float(g["curv_error_max"])) < CURV_TOL, name
        if name == "oval_n2000_kappa":
            assert info[0]["n_active_kappa"] > 0 and abs(info[0]["kappa_max"] - kb) < 1e-9
    print("round-4 N=2000 dense goldens: max |alpha - dense oracle| per fixture: %s" % {k: "%.1e" % v for k, v in worst.items()})


def test_oval_n2000_iqp_end_state_against_golden(gpu_engine):
    """BASELINE config 3 IS mincurv_iqp: the END STATE of the whole iqp_handler chain at N = 2000 (three passes, N = 2000 ->
    2003 -> 2002: re-sampling, width carry-over, re-spline, damping 1/3 and 2/3) against the committed output of the oracle's
    chain (dense re-linearisation every pass, ~2 min of CPU): through the drop-in function, through the host-glue batch driver,
    through the device-resident driver cold, and through the device-resident driver with warm-started passes."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler as iq
    g = _golden_n2000()
    A = tph.calc_splines.build_les_matrix(2000, g["scaling"])
    step = float(g["stepsize_interp"])
    outs = [tph.iqp_handler.iqp_handler(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], A=A, kappa_bound=0.12, w_veh=3.4,
                                        print_debug=False, plot_debug=False, stepsize_interp=step, iters_min=3,
                                        curv_error_allowed=0.01)]
    trk = [dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"])]
    for kw in (dict(device_resident=False), dict(device_resident=True, warm_start=False), dict(device_resident=True, warm_start=True)):
        stt = {}
        outs.append(iq.iqp_handler_batch(trk, 0.12, 3.4, step, 3, 0.01, engine=gpu_engine, stats=stt, **kw)[0])
        assert stt["rounds"] == len(g["iqp_n"]) == 3
    for a, ref_out, nv_out in outs:
        assert a.shape == g["iqp_alpha"].shape == (int(g["iqp_n"][-1]),)
        assert np.max(np.abs(a - g["iqp_alpha"])) < ALPHA_TOL, float(np.max(np.abs(a - g["iqp_alpha"])))
        assert np.max(np.abs(ref_out - g["iqp_reftrack"])) < 1e-6
        assert np.max(np.abs(nv_out - g["iqp_normvec"])) < 1e-8


def test_many_active_curvature_rows_against_dense_gi(gpu_engine, golden):
    """quadprog carries any number of active curvature rows [REF params/racecar.ini:49 curvlim]; so must the engine.  The handling
    track with kappa_bound tightened until 40 / 51 curvature rows are active at the optimum (dense Goldfarb-Idnani, all 4N rows),
    and Berlin at 0.07 (11 rows on a 776-point ring): alpha at the north_star tolerance, the same number of active rows, the
    bound met to round-off, refinement run on the curvature-row working set; and just beyond feasibility: status 5 ->
    ValueError("constraints are inconsistent, no solution") like quadprog."""
    from oracle import qp_ref, tph_ref
    cases = [("handling_track", 0.0495), ("handling_track", 0.048), ("berlin_2018", 0.07)]
    probs, want = [], []
    for name, kb in cases:
        g = golden[name]
        n = g["reftrack"].shape[0]
        A = tph.calc_splines.build_les_matrix(n, g["scaling"])
        info = {}
        a_ref, err_ref = tph_ref.opt_min_curv(g["reftrack"], g["normvec"], A, kb, 3.4,
                                              solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
        want.append((a_ref, err_ref, int(np.sum(info["lagr"][2 * n:] > 0))))
        probs.append(dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=kb, w_veh=3.4))
    assert want[0][2] >= 40 and want[1][2] >= 50
    al, curv, st, info = gpu_engine.solve_batch(probs)
    for k, (a_ref, err_ref, nk) in enumerate(want):
        assert st[k] == 0, (cases[k], st[k])
        assert np.max(np.abs(al[k] - a_ref)) < ALPHA_TOL, (cases[k], float(np.max(np.abs(al[k] - a_ref))))
        assert abs(curv[k] - err_ref) < CURV_TOL
        assert info[k]["n_active_kappa"] == nk, (cases[k], info[k]["n_active_kappa"], nk)
        assert abs(info[k]["kappa_max"] - cases[k][1]) < 1e-9 and info[k]["refine_rounds"] >= 1
    g = golden["handling_track"]
    A = tph.calc_splines.build_les_matrix(g["reftrack"].shape[0], g["scaling"])
    with pytest.raises(ValueError, match="inconsistent"):
        tph_ref.opt_min_curv(g["reftrack"], g["normvec"], A, 0.04, 3.4)
    with pytest.raises(ValueError, match="constraints are inconsistent, no solution"):
        tph.opt_min_curv.opt_min_curv(g["reftrack"], g["normvec"], A, 0.04, 3.4)


def test_poisoned_workspaces_and_lds_bitwise(gpu_engine, golden, monkeypatch):
    """MCQ_POISON=1 (workspaces, staging buffers and the solver kernel's LDS start out as NaN patterns): the golden tracks, a
    curvature-row case, a 64-problem full-size batch and a warm-started IQP come out BITWISE as on the unpoisoned engine -- nothing
    the kernels read was left over from whatever ran on the CU before (round 2: the fused forward substitution's backward sweep)."""
    monkeypatch.setenv("MCQ_POISON", "1")
    eng = engine.Engine(0)
    try:
        probs = [_problem(golden[k]) for k in golden] + [dict(_problem(golden["rounded_rectangle"]), kappa_bound=0.10)]
        ref, nv, sc = synthetic.oval_batch(64, 2000, first=5)
        probs += [dict(reftrack=ref[k], normvec=nv[k], scaling=sc[k], kappa_bound=0.12, w_veh=2.0) for k in range(64)]
        a1, c1, s1, _ = eng.solve_batch(probs)
        a0, c0, s0, _ = gpu_engine.solve_batch(probs)
        assert list(s1) == list(s0) and np.array_equal(c1, c0)
        assert all(np.array_equal(x, y) for x, y in zip(a1, a0))
        for k, name in enumerate(golden):
            assert s1[k] == 0 and np.max(np.abs(a1[k] - golden[name]["alpha"])) < ALPHA_TOL
        tracks = [dict(reftrack=ref[k].copy(), normvectors=nv[k], scaling=sc[k]) for k in range(8)]
        o1 = tph.iqp_handler.iqp_handler_batch(tracks, 0.12, 2.0, 3.0, 4, 0.01, engine=eng, device_resident=True, warm_start=True)
        o0 = tph.iqp_handler.iqp_handler_batch(tracks, 0.12, 2.0, 3.0, 4, 0.01, engine=gpu_engine, device_resident=True, warm_start=True)
        assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(o1, o0))
    finally:
        eng.close()



def test_config4_full_size_lap_time_matrix(gpu_engine, golden):
    """BASELINE config 4 at FULL size on one GPU (only a miniature was tested): the lap-time matrix of 16 384 variants =
    4 reference tracks x 64 vehicle widths x 64 (gg-scale, top-speed) vehicles -- 256 QPs in one ragged launch, their racelines on
    the device, 16 384 ragged velocity profiles.  A sample of the QPs (one per track, spread over the width grid) against the live
    dense oracle, a sample of the lap times / profiles against oracle/vel_ref.py on the host chain's raceline, every variant
    finite and the matrix monotone where it must be (more grip or more top speed never costs lap time)."""
    import bench
    from oracle import tph_ref, vel_ref
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import create_raceline as cr, calc_head_curv_an as ch
    wl = bench.config4_workload(0, 1)
    assert wl["n_total"] == 16384 and len(wl["qps"]) == 256
    al, curv, st, info = gpu_engine.solve_batch(wl["qps"])
    assert np.all(st == 0)
    race = gpu_engine.raceline_batch([p["reftrack"] for p in wl["qps"]], [p["normvec"] for p in wl["qps"]], al, 2.0)
    assert np.all(race["status"] == 0)
    vx, lap = gpu_engine.vel_profile_batch(race["kappa"], race["el_lengths"], wl["ggv"], wl["axm"], 0.75, 1200.0, wl["tops"], 1.0,
                                           track_of=wl["track_of"], n_of_track=race["m"])
    assert lap.shape == (16384,) and np.all(np.isfinite(lap)) and np.all(lap > 5.0)
    # ---- QPs against the dense oracle: one per track, at different points of the width grid
    worst = 0.0
    for q in (5, 64 + 40, 128 + 63, 192 + 20):
        p = wl["qps"][q]
        A = tph.calc_splines.build_les_matrix(p["reftrack"].shape[0], p["scaling"])
        a_ref, err_ref = tph_ref.opt_min_curv(p["reftrack"], p["normvec"], A, p["kappa_bound"], p["w_veh"])
        worst = max(worst, float(np.max(np.abs(al[q] - a_ref))))
        assert np.max(np.abs(al[q] - a_ref)) < ALPHA_TOL, q
        assert abs(curv[q] - err_ref) < CURV_TOL, q
    # ---- lap times / profiles against the oracle's velocity profile on the host chain's raceline
    rng = np.random.default_rng(5)
    for j in [int(k) for k in rng.choice(16384, size=12, replace=False)]:
        q = int(wl["track_of"][j])
        p = wl["qps"][q]
        out = cr.create_raceline(refline=p["reftrack"][:, :2], normvectors=p["normvec"], alpha=al[q], stepsize_interp=2.0)
        _, kap = ch.calc_head_curv_an(coeffs_x=out[2], coeffs_y=out[3], ind_spls=out[4], t_spls=out[5])
        el = out[8]
        vx_o = vel_ref.calc_vel_profile(ax_max_machines=wl["axm"][j], kappa=kap, el_lengths=el, closed=True, drag_coeff=0.75, m_veh=1200.0,
                                        ggv=wl["ggv"][j], v_max=wl["tops"][j], dyn_model_exp=1.0)
        assert int(race["m"][q]) == kap.size
        assert np.max(np.abs(vx[j, :kap.size] - vx_o)) < 1e-7, j
        assert abs(lap[j] - vel_ref.lap_time_stable(vx_o, el)) < 1e-7, j
    # ---- structure of the matrix: per QP the 64 vehicles are an 8 x 8 (gg-scale, top-speed) grid
    m = lap.reshape(256, 8, 8)                       # [qp][top-speed index][gg-scale index]
    assert np.all(np.diff(m, axis=2) <= 1e-9)        # more grip: never slower
    # (a higher top speed is NOT monotone in tph's profile -- the acceleration-phase gating of its sweeps can cost a few ms when v_max
    #  moves a phase start -- so only the ends of the top-speed axis are compared, with that slack)
    assert np.all(m[:, -1, :] <= m[:, 0, :] + 0.05)
    print("config 4 full size: 256 QPs (worst sampled |alpha - dense oracle| %.2e m), 16384 lap times %.2f ... %.2f s" % (worst, lap.min(), lap.max()))


def test_bench_force_collective_initialises_rccl():
    """The N > 1 path of bench.py on a 1-GPU box: `--force-collective` brings up RCCL THROUGH THE C ABI (mcq_comm_init: the id travels
    over the gloo rendezvous; ncclAllGather on the engine's comm stream, round 4 -- no torch on the GPU), runs the all-gather of alpha
    inside the timed region and checks the own shard -- so every round's GPU suite exercises the collective's start-up, its ordering
    behind the solves, and the line's multi-GPU fields."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MCQ_LIB"):
        env.pop(k, None)
    env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", "29577"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-collective", "--steps", "2", "--warmup", "1", "--batch", "64",
                          "--no-extras"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    c = rec["config"]
    assert c["collective"].startswith("1 all-gather of alpha per step: ncclAllGather (RCCL) through the C ABI")
    assert c["ranks_seen"] == 1 and rec["n_gpus"] == 1
    assert c["allgather_ms"] is not None and np.isfinite(c["allgather_ms"]) and 0.0 < c["allgather_ms"] < 50.0
    assert c["failed_problems"] == 0 and rec["value"] > 0


def test_engine_collective_and_sharded_solve_on_one_rank(gpu_engine, golden):
    """The C ABI's collective on its own (mcq_comm_*: RCCL loaded with dlopen, one rank) and parallel.solve_sharded's device-resident
    path through it (that path had no GPU test): gathered buffers equal the send buffers bit for bit, for every dtype; the
    sharded solve returns what the host-buffer entry returns; a batch without normals is accepted (derived on the device)."""
    from global_racetrajectory_optimization_amd import parallel
    eng = engine.Engine(0)           # a handle of its own: the session's engine keeps no communicator
    try:
        eng.comm_init(0, 1, eng.comm_unique_id())
        assert eng.comm_world() == (0, 1)
        rng = np.random.default_rng(5)
        for np_dt, code in ((np.float64, eng.DT_F64), (np.float32, eng.DT_F32), (np.int32, eng.DT_I32)):
            a = (rng.standard_normal(70001) * 1000).astype(np_dt)
            d_s, d_r = eng.alloc(a.nbytes), eng.alloc(a.nbytes)
            eng.upload(d_s, a)
            eng.comm_allgather(d_s, d_r, a.size, code)
            assert eng.comm_wait(0) >= 0.0
            assert np.array_equal(eng.download(d_r, a.shape, np_dt), a)
            eng.free(d_s)
            eng.free(d_r)
        probs = [dict(reftrack=golden[t]["reftrack"], normvec=golden[t]["normvec"], scaling=golden[t]["scaling"], kappa_bound=0.12, w_veh=3.4)
                 for t in ("rounded_rectangle", "handling_track", "modena_2019")]
        a_s, c_s, s_s = parallel.solve_sharded(probs, eng)
        a_h, c_h, s_h, _ = gpu_engine.solve_batch(probs)
        assert list(s_s) == list(s_h) == [0, 0, 0] and np.array_equal(c_s, c_h)
        assert all(np.array_equal(x, y) for x, y in zip(a_s, a_h))
        bare = [dict(p, normvec=None, scaling=None) for p in probs]
        a_b, _, s_b = parallel.solve_sharded(bare, eng)
        assert list(s_b) == [0, 0, 0] and max(float(np.max(np.abs(x - y))) for x, y in zip(a_b, a_h)) < 1e-6
        with pytest.raises(ValueError, match="for all problems or for none"):
            parallel.solve_sharded([probs[0], bare[1]], eng)
        eng.comm_destroy()
        with pytest.raises(engine.EngineError):
            eng.comm_world()
        # the handle stays usable after comm_destroy -- a download must not wait on the destroyed events of the last gather --
        # and takes a new communicator
        d_x = eng.alloc(80)
        eng.upload(d_x, np.arange(10.0))
        assert np.array_equal(eng.download(d_x, (10,), np.float64), np.arange(10.0))
        a_h2, _, s_h2, _ = eng.solve_batch(probs[:1])
        assert s_h2[0] == 0 and np.array_equal(a_h2[0], a_h[0])
        eng.comm_init(0, 1, eng.comm_unique_id())
        d_y = eng.alloc(80)
        eng.comm_allgather(d_x, d_y, 10, eng.DT_F64)
        assert np.array_equal(eng.download(d_y, (10,), np.float64), np.arange(10.0))
        eng.free(d_x)
        eng.free(d_y)
        eng.comm_destroy()
    finally:
        eng.close()


def test_curvature_row_overflow_slots_against_dense_gi(gpu_engine):
    """quadprog carries any number of active curvature rows [REF params/racecar.ini:49 curvlim]; the engine's
    LDS-resident Schur path holds MCQ_KMAX = 120.  Stadium tracks whose optimal line has its curvature on a plateau, bound just
    below it: 134 rows at n = 360 and about twice that at n = 720, plus three copies in one launch (each claims its own overflow
    slot) next to an ordinary problem -- all against the dense Goldfarb-Idnani oracle with all 4N rows."""
    from oracle import qp_ref, tph_ref
    from test_emu_kernels import stadium_problem
    probs, want = [], []
    for n, kb in ((360, 0.0223), (720, 0.0223)):
        ref, nv, A, sc, kb = stadium_problem(n, kb)
        info = {}
        a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, kb, 2.0, solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
        nk = int(np.sum(info["lagr"][2 * n:] > 0))
        for _ in range(3 if n == 360 else 1):
            probs.append(dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=kb, w_veh=2.0))
            want.append((a_ref, err_ref, nk))
    assert want[0][2] > 120 and want[-1][2] > 200
    al, curv, st, info = gpu_engine.solve_batch(probs)
    for k, (a_ref, err_ref, nk) in enumerate(want):
        assert st[k] == 0, (k, st[k])
        assert info[k]["n_active_kappa"] == nk, (k, info[k]["n_active_kappa"], nk)
        assert np.max(np.abs(al[k] - a_ref)) < ALPHA_TOL, (k, float(np.max(np.abs(al[k] - a_ref))))
        assert abs(curv[k] - err_ref) < CURV_TOL
    print("curvature-row overflow: %s active rows, max |alpha - dense GI| = %.2e m"
          % ([w[2] for w in want], max(float(np.max(np.abs(al[k] - w[0]))) for k, w in enumerate(want))))
    # more such problems in ONE launch than the handle has overflow slots (8): which of them get a slot depends on the order the GPU
    # schedules workgroups in -- the ones left out go through the Goldfarb-Idnani path inside the same kernel (round 5; rounds 3-4:
    # MCQ_KAPPA_NO_SLOT, re-launched by the host-buffer entries only), whose final round from the same working set is the block-pivoting
    # phase's own: every copy comes back with status 0 and the same vertex, on the host-buffer entry and on the device entry alike
    many = [probs[0]] * 11 + [dict(reftrack=probs[0]["reftrack"], normvec=probs[0]["normvec"], scaling=probs[0]["scaling"], kappa_bound=0.5,
                                  w_veh=2.0)]
    al2, curv2, st2, info2 = gpu_engine.solve_batch(many)
    assert np.all(st2 == 0)
    for k in range(11):
        # round 6: ONE route per problem -- every working set beyond MCQ_KMAX rows goes through the Goldfarb-Idnani path, whatever
        # else is in the launch -- so the copies are BITWISE the single solve again (rounds 3-5: 1e-9, eight of them through overflow slots)
        assert np.array_equal(al2[k], al[0]) and curv2[k] == curv[0] and info2[k]["n_active_kappa"] == want[0][2], k
        assert info2[k]["second_attempt"] & 4
    print("11 copies, 8 overflow slots: Goldfarb-Idnani path for %d of them, bitwise equal to the single solve: %d of 11" % (
        sum(1 for i in info2[:11] if i["second_attempt"] & 4), sum(1 for k in range(11) if np.array_equal(al2[k], al[0]))))
    n0 = probs[0]["reftrack"].shape[0]
    d_ref, d_nv, d_sc = (gpu_engine.alloc(8 * 11 * n0 * w) for w in (4, 2, 1))
    d_al, d_cu, d_st = gpu_engine.alloc(8 * 11 * n0), gpu_engine.alloc(8 * 11), gpu_engine.alloc(4 * 11)
    gpu_engine.upload(d_ref, np.tile(probs[0]["reftrack"], (11, 1, 1)))
    gpu_engine.upload(d_nv, np.tile(probs[0]["normvec"], (11, 1, 1)))
    gpu_engine.upload(d_sc, np.tile(probs[0]["scaling"], (11, 1)))
    gpu_engine.solve_device(11, n0, d_ref, d_nv, d_sc, probs[0]["kappa_bound"], 2.0, d_al, d_cu, d_st)
    st3 = gpu_engine.download(d_st, (11,), np.int32)
    assert np.all(st3 == 0), st3
    al3 = gpu_engine.download(d_al, (11, n0), np.float64)
    assert np.array_equal(al3, np.tile(al[0], (11, 1)))            # ... on the device entry too
    for p_ in (d_ref, d_nv, d_sc, d_al, d_cu, d_st):
        gpu_engine.free(p_)

