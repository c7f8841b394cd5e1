"""CPU execution of the UNCHANGED HIP kernel sources through the SIMT interpreter in tests/emu (test infrastructure
only -- see the header of tests/emu/include/hip/hip_runtime.h).  Exercises every kernel's index arithmetic, barrier
placement and wave-level data flow against the golden vectors without a GPU; the `-m gpu` suite repeats the parity
checks on real hardware through the real libmcq.so."""
import os

import numpy as np
import pytest

from global_racetrajectory_optimization_amd import engine
from oracle import tph_ref


@pytest.fixture(scope="module")
def emu(emu_lib):
    eng = engine.Engine(0, lib_path=emu_lib)
    yield eng
    eng.close()


def _problem(g):
    return dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=float(g["kappa_bound"]),
                w_veh=float(g["w_veh"]))


def test_rounded_rectangle_matches_golden(emu, golden):
    g = golden["rounded_rectangle"]
    al, curv, st, info = emu.solve_batch([_problem(g)])
    assert st[0] == 0
    assert np.max(np.abs(al[0] - g["alpha"])) < 1e-9
    assert abs(curv[0] - float(g["curv_error_max"])) < 1e-10
    assert info[0]["n_active_box"] == 15 and info[0]["kkt_res"] < 1e-10
    assert 8 <= info[0]["ipm_iters"] <= 16 and info[0]["as_iters"] >= 1        # (the interpreter reported 0 iterations until round 5: a missing barrier)


def _small_track(n, seed):
    rng = np.random.default_rng(seed)
    th = np.linspace(0.0, 2 * np.pi, n, endpoint=False)
    r = 40.0 + 6.0 * np.sin(3 * th + rng.uniform(0, 6)) + 3.0 * np.cos(5 * th + rng.uniform(0, 6))
    xy = np.column_stack((r * np.cos(th), r * np.sin(th)))
    path_cl = np.vstack((xy, xy[0]))
    _, _, A, nv = tph_ref.calc_splines(path_cl)
    w = 3.0 + rng.uniform(0.0, 1.5, size=(n, 2))
    ref = np.column_stack((xy, w))
    idx = np.arange(n - 1)
    sc = np.empty(n)
    sc[:-1] = -A[4 * idx + 2, 4 * idx + 5]
    sc[-1] = A[4 * n - 2, 1]
    return ref, nv, A, sc


@pytest.mark.parametrize("n", [7, 20, 33, 64, 70])
def test_small_and_ragged_rings_match_dense_oracle(emu, n):
    """Short rings exercise the image folding of the periodic Green's function, the asymmetric band for even n and the
    small-n border/band dimension logic; a ragged batch exercises the per-problem n path."""
    ref, nv, A, sc = _small_track(n, seed=n)
    a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, 0.5, 2.0)
    ref2, nv2, A2, sc2 = _small_track(n + 3, seed=n + 100)
    a_ref2, err_ref2 = tph_ref.opt_min_curv(ref2, nv2, A2, 0.5, 2.0)
    al, curv, st, _ = emu.solve_batch([dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=0.5, w_veh=2.0),
                                       dict(reftrack=ref2, normvec=nv2, scaling=sc2, kappa_bound=0.5, w_veh=2.0)])
    assert list(st) == [0, 0]
    assert np.max(np.abs(al[0] - a_ref)) < 1e-8
    assert np.max(np.abs(al[1] - a_ref2)) < 1e-8
    assert abs(curv[0] - err_ref) < 1e-9 and abs(curv[1] - err_ref2) < 1e-9


def test_status_codes(emu, golden):
    g = golden["rounded_rectangle"]
    p = _problem(g)
    narrow = dict(p)
    narrow["reftrack"] = g["reftrack"].copy()
    narrow["reftrack"][5, 2:] = 1.0            # w_r + w_l < w_veh
    bad = dict(p)
    bad["reftrack"] = g["reftrack"].copy()
    bad["reftrack"][3, 0] = np.nan
    al, curv, st, _ = emu.solve_batch([narrow, bad])
    assert st[0] == engine.STATUS_INFEASIBLE and st[1] == engine.STATUS_BAD_INPUT


def test_curvature_rows_active_and_infeasible(emu):
    """Curvature-bound rows of the QP: active at the optimum -> exact vertex through the Schur
    complement path; impossible to satisfy -> status 5 (quadprog: 'constraints are inconsistent, no solution')."""
    from oracle import qp_ref
    ref, nv, A, sc = _small_track(40, seed=5)
    a_box, _, I = tph_ref.opt_min_curv(ref, nv, A, 10.0, 2.0, return_internals=True)
    kmax = float(np.max(np.abs(I["k_ref"] + I["E"] @ a_box)))
    kb = 0.9 * kmax
    info = {}
    a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, kb, 2.0, solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
    n_act_kappa = int(np.sum(info["lagr"][2 * 40:] > 0))
    assert n_act_kappa >= 1
    with pytest.raises(ValueError, match="inconsistent"):
        tph_ref.opt_min_curv(ref, nv, A, 1e-4, 2.0)
    al, curv, st, inf = emu.solve_batch([dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=kb, w_veh=2.0),
                                         dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=1e-4, w_veh=2.0)],
                                        max_ipm_iter=30)       # the infeasible one runs into the cap: keep the interpreter's bill down
    assert st[0] == 0 and st[1] == engine.STATUS_KAPPA_INFEASIBLE
    assert inf[0]["n_active_kappa"] == n_act_kappa
    assert np.max(np.abs(al[0] - a_ref)) < 1e-8
    assert abs(curv[0] - err_ref) < 1e-9
    assert abs(inf[0]["kappa_max"] - kb) < 1e-9


def _relin_device(eng, tracks, alphas, alpha_scale, stepsize, nmax):
    """Drives mcq_relinearise_device with plain host arrays (the interpreter's "device" memory is host memory)."""
    bsz = len(tracks)
    n_in = np.array([t[0].shape[0] for t in tracks], dtype=np.int32)
    ref_in = np.zeros((bsz, nmax, 4))
    nv_in = np.zeros((bsz, nmax, 2))
    al = np.zeros((bsz, nmax))
    for k, (ref, nv) in enumerate(tracks):
        ref_in[k, :n_in[k]] = ref
        nv_in[k, :n_in[k]] = nv
        al[k, :n_in[k]] = alphas[k]
    ref_out = np.zeros_like(ref_in)
    nv_out = np.zeros_like(nv_in)
    n_out = np.zeros(bsz, dtype=np.int32)
    st = np.full(bsz, -1, dtype=np.int32)
    eng.relinearise_device(bsz, nmax, n_in.ctypes.data, ref_in.ctypes.data, nv_in.ctypes.data, al.ctypes.data, None,
                           alpha_scale, stepsize, ref_out.ctypes.data, nv_out.ctypes.data, n_out.ctypes.data,
                           st.ctypes.data)
    eng.sync()
    return ref_out, nv_out, n_out, st


def test_relinearise_kernel_matches_host_glue(emu, golden):
    """Row f-1: the device-side IQP glue (raceline, re-sampling, widths, normals) against the host chain
    create_raceline -> interp_track_widths -> calc_splines(use_dist_scaling=False), long and short rings in one launch."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler as iqp
    g = golden["rounded_rectangle"]
    ref_s, nv_s, _, _ = _small_track(40, 3)                      # short ring: exact periodic spline kernel
    rng = np.random.default_rng(0)
    tracks = [(g["reftrack"], g["normvec"]), (ref_s, nv_s)]
    alphas = [g["alpha"], 0.3 * rng.uniform(-1.0, 1.0, size=40)]
    for scale, step in ((1.0 / 3.0, 3.0), (1.0, 2.0)):
        ref_d, nv_d, n_d, st = _relin_device(emu, tracks, alphas, scale, step, nmax=512)
        for k, (ref, nv) in enumerate(tracks):
            ref_h, nv_h = iqp._relinearise(np.array(ref), np.array(nv), scale * alphas[k], step)
            assert st[k] == 0
            assert n_d[k] == ref_h.shape[0]
            m = n_d[k]
            assert np.max(np.abs(ref_d[k, :m] - ref_h)) < 1e-9
            assert np.max(np.abs(nv_d[k, :m] - nv_h)) < 1e-9
    # a ring that does not fit the output stride is reported, not truncated
    _, _, _, st = _relin_device(emu, tracks[:1], alphas[:1], 1.0, 0.5, nmax=512)
    assert st[0] == engine.STATUS_BAD_INPUT


def test_prep_on_device_matches_calc_splines(emu, golden):
    """Row f-2: normals and spline scalings of the closed distance-scaled spline computed by the assembly kernel
    (mcq_prep_device) against tph.calc_splines as prep_track calls it; and a solve that is given no normals / scalings
    must return what the solve with the host-side ones returns."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs
    g = golden["rounded_rectangle"]
    ref_s, _, _, _ = _small_track(40, 5)
    refs = [g["reftrack"], ref_s]
    nvs, scs = emu.prep_batch(refs)
    for ref, nv_d, sc_d in zip(refs, nvs, scs):
        path_cl = np.vstack((ref[:, :2], ref[0, :2]))
        _, _, A, nv_h = cs.calc_splines(path=path_cl)
        assert np.max(np.abs(nv_d - nv_h)) < 1e-10
        assert np.max(np.abs(sc_d - cs.scalings_from_les_matrix(A))) < 1e-12
    al_d, curv_d, st_d, _ = emu.solve_batch([dict(reftrack=g["reftrack"], normvec=None, scaling=None,
                                                  kappa_bound=float(g["kappa_bound"]), w_veh=float(g["w_veh"]))])
    assert st_d[0] == 0
    assert np.max(np.abs(al_d[0] - g["alpha"])) < 1e-8
    assert abs(curv_d[0] - float(g["curv_error_max"])) < 1e-9


def _raceline_kappa_el(g, stepsize=3.0):
    """Curvature and element lengths of the golden raceline of a track, the way main_globaltraj.py derives them."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import create_raceline as cr, calc_head_curv_an as ch
    out = cr.create_raceline(refline=g["reftrack"][:, :2], normvectors=g["normvec"], alpha=g["alpha"], stepsize_interp=stepsize)
    _, _, cx, cy, inds, tvals, _, _, el_cl = out
    _, kappa = ch.calc_head_curv_an(coeffs_x=cx, coeffs_y=cy, ind_spls=inds, t_spls=tvals)
    return kappa, el_cl


def _vehicle_variants():
    """ggv / machine tables in the format of inputs/veh_dyn_info [REF ggv.csv, ax_max_machines.csv], scaled like the
    lap-time-matrix sweep scales them [REF main_globaltraj.py:442-496]."""
    v = np.arange(0.0, 72.1, 4.0)
    ggv0 = np.column_stack((v, np.full(v.size, 12.0), np.full(v.size, 12.0)))
    axm = np.column_stack((v, np.interp(v, [0.0, 20.0, 72.0], [5.3, 5.3, 1.2])))
    variants = []
    for scale, vmax in ((1.0, 70.0), (0.6, 50.0), (0.35, 38.0)):
        gg = ggv0.copy()
        gg[:, 1:] *= scale
        variants.append((gg, axm, 0.75, 1200.0, vmax))
    return variants


def _vehicle_variants_speed_dependent():
    """A ggv diagram whose limits change with speed (downforce car: more lateral grip at speed; and one that loses grip), top
    speeds BETWEEN the grid points and below the end of the tables: the fixed-point iteration of the lateral limit runs more
    than one round and the rows above v_max enter the interpolation (nothing is truncated upstream)."""
    v = np.arange(0.0, 72.1, 4.0)
    axm = np.column_stack((v, np.interp(v, [0.0, 20.0, 72.0], [5.3, 5.3, 1.2])))
    up = np.column_stack((v, 10.0 + 0.05 * v, 9.0 + 0.12 * v))
    down = np.column_stack((v, 12.0 - 0.04 * v, 13.0 - 0.1 * v))
    return [(up, axm, 0.75, 1200.0, 53.7), (down, axm, 0.6, 900.0, 45.1), (up, axm, 0.75, 1200.0, 30.3)]


def _check_vel_profiles(eng, kappa, el, var, dyn_model_exp, tol=1e-9, mu=None, filt_window=None):
    """Device velocity profiles / lap times of `var` on one raceline against oracle/vel_ref.py (the upstream algorithm
    restated: acceleration-phase gating, backward look-ahead, no ggv truncation) and against the product's host shim."""
    from oracle import vel_ref
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_vel_profile as cv
    vx_d, lt_d = eng.vel_profile_batch(kappa[None, :], el[None, :], np.stack([v[0] for v in var]), np.stack([v[1] for v in var]),
                                       [v[2] for v in var], [v[3] for v in var], [v[4] for v in var], dyn_model_exp=dyn_model_exp,
                                       track_of=np.zeros(len(var), dtype=np.int32), mu=None if mu is None else mu[None, :],
                                       filt_window=filt_window)
    rounds = []
    for k, (gg, axm, drag, mass, vmax) in enumerate(var):
        info = {}
        vx_o = vel_ref.calc_vel_profile(ax_max_machines=axm, kappa=kappa, el_lengths=el, closed=True, drag_coeff=drag, m_veh=mass,
                                        ggv=gg, v_max=vmax, dyn_model_exp=dyn_model_exp, mu=mu, filt_window=filt_window, info=info)
        rounds.append(info["lateral_rounds"])
        ax_o = vel_ref.calc_ax_profile(np.append(vx_o, vx_o[0]), el)
        t_o = vel_ref.calc_t_profile(vx_o, el, ax_profile=ax_o)
        assert np.max(np.abs(vx_d[k] - vx_o)) < tol, k
        assert abs(lt_d[k] - vel_ref.lap_time_stable(vx_o, el)) < tol, k
        assert abs(lt_d[k] - t_o[-1]) < 0.5                 # upstream's own expression: noisy as a -> 0 (see vel_ref.lap_time_stable)
        vx_h = cv.calc_vel_profile(ggv=gg, ax_max_machines=axm, v_max=vmax, kappa=kappa, el_lengths=el, closed=True,
                                   filt_window=filt_window, dyn_model_exp=dyn_model_exp, drag_coeff=drag, m_veh=mass, mu=mu)
        assert np.max(np.abs(vx_h - vx_o)) < 1e-12, k      # the shim main_globaltraj.py calls: same numbers
    return rounds


def test_velocity_profile_kernel_matches_oracle(emu, golden):
    """Row f-3: the batched ggv velocity profile + lap time kernel against the ORACLE's restatement of tph.calc_vel_profile ->
    calc_ax_profile -> calc_t_profile: the reference's constant ggv scaled like the lap-time matrix scales it, and
    speed-dependent diagrams with top speeds between the grid points; friction-ellipse exponents 1 and 2; uniform element
    lengths (what create_raceline hands on) and non-uniform ones (backward step p -> p-1 uses el[p], as upstream)."""
    kappa, el = _raceline_kappa_el(golden["rounded_rectangle"])
    _check_vel_profiles(emu, kappa, el, _vehicle_variants(), 1.0)
    rounds = _check_vel_profiles(emu, kappa, el, _vehicle_variants_speed_dependent(), 2.0)
    assert max(rounds) > 1
    el_nu = el * (1.0 + 0.3 * np.sin(0.7 * np.arange(el.size)))
    _check_vel_profiles(emu, kappa, el_nu, _vehicle_variants_speed_dependent(), 1.0)
    # a straight (kappa == 0 exactly: infinite radius; numpy's NaN-propagating max keeps upstream iterating all 100 rounds)
    kz = kappa.copy()
    kz[5:9] = 0.0
    rounds = _check_vel_profiles(emu, kz, el, _vehicle_variants_speed_dependent()[:2], 1.0)
    assert rounds == [100, 100]


def test_velocity_profile_filter_window_and_friction_map(emu, golden):
    """The options of tph.calc_vel_profile the reference reads from its parameter file or leaves at their defaults, on the device
    (mcq_vel_profile_device_opts): vel_profile_conv_filt_window [REF params/racecar.ini:54-57, main_globaltraj.py:407] -- a closed
    moving average over the finished profile, lap time from the filtered profile -- and a friction coefficient per waypoint (the
    first estimate of the lateral limit uses its mean, as upstream); both together; an even window raises as tph does."""
    kappa, el = _raceline_kappa_el(golden["rounded_rectangle"])
    n = kappa.size
    mu = 0.9 + 0.2 * np.cos(2.0 * np.pi * np.arange(n) / n * 3.0)
    _check_vel_profiles(emu, kappa, el, _vehicle_variants()[:3], 1.0, filt_window=5)
    _check_vel_profiles(emu, kappa, el, _vehicle_variants_speed_dependent()[:2], 2.0, mu=mu)
    _check_vel_profiles(emu, kappa, el, _vehicle_variants_speed_dependent()[:2], 1.0, mu=mu, filt_window=3)
    var = _vehicle_variants()[:1]
    with pytest.raises(RuntimeError, match="must be odd"):
        emu.vel_profile_batch(kappa[None, :], el[None, :], np.stack([v[0] for v in var]), np.stack([v[1] for v in var]),
                              [v[2] for v in var], [v[3] for v in var], [v[4] for v in var], filt_window=4)
    with pytest.raises(ValueError, match="wider than the shortest profile"):       # (the kernel's NaN flag used to come back silently)
        emu.vel_profile_batch(kappa[None, :], el[None, :], np.stack([v[0] for v in var]), np.stack([v[1] for v in var]),
                              [v[2] for v in var], [v[3] for v in var], [v[4] for v in var], filt_window=2 * (n // 2) + 3)


def test_velocity_profile_table_range_errors(emu, golden):
    """ggv / machine tables that end below v_max: tph raises RuntimeError; so does the host wrapper, and the kernel flags NaN."""
    from oracle import vel_ref
    kappa, el = _raceline_kappa_el(golden["rounded_rectangle"])
    gg, axm, drag, mass, _ = _vehicle_variants()[0]
    with pytest.raises(RuntimeError, match="ggv has to cover"):
        vel_ref.calc_vel_profile(axm, kappa, el, True, drag, mass, ggv=gg, v_max=80.0)
    with pytest.raises(RuntimeError, match="ggv has to cover"):
        emu.vel_profile_batch(kappa[None, :], el[None, :], gg[None], axm[None], [drag], [mass], [80.0])
    with pytest.raises(RuntimeError, match="ax_max_machines has to cover"):
        emu.vel_profile_batch(kappa[None, :], el[None, :], np.vstack((gg, [[90.0, 12.0, 12.0]]))[None], axm[None], [drag], [mass], [80.0])


@pytest.mark.parametrize("n", [7, 33, 70])
def test_shortest_path_objective_matches_dense_oracle(emu, n):
    """Row f-4 through the unchanged kernels: the cyclic tridiagonal H as two vectors, its scalar elimination (Sherman-Morrison
    for the ring, per-thread blocks + cyclic reduction of the separators; n = 7 / 33 / 70: blocks of one row, i.e. the reduction alone),
    the H x + f gradient, the 1 mm clipping of the deviations and a ragged batch; against the dense Goldfarb-Idnani oracle."""
    ref, nv, _, _ = _small_track(n, seed=n)
    ref2, nv2, _, _ = _small_track(n + 3, seed=n + 100)
    ref2[: n // 2, 2] = 0.9                  # w_r - w_veh/2 < 0: clipped to 0.001
    a1 = tph_ref.opt_shortest_path(ref, nv, 2.0)
    a2 = tph_ref.opt_shortest_path(ref2, nv2, 2.0)
    al, curv, st, info = emu.solve_batch([dict(reftrack=ref, normvec=nv, scaling=None, kappa_bound=1.0, w_veh=2.0),
                                          dict(reftrack=ref2, normvec=nv2, scaling=None, kappa_bound=1.0, w_veh=2.0)],
                                         objective=engine.OBJ_SHORTEST_PATH)
    assert list(st) == [0, 0] and np.all(curv == 0.0)
    assert np.max(np.abs(al[0] - a1)) < 1e-9
    assert np.max(np.abs(al[1] - a2)) < 1e-9
    assert np.all(al[1][: n // 2] <= 0.001 + 1e-15)
    assert info[0]["kkt_res"] < 1e-10 and info[0]["n_active_box"] > 0


def _kkt_box_certificate(H_mul, f, lo, hi, x, tol):
    """KKT conditions of  min 1/2 x'Hx + f'x, lo <= x <= hi  checked on the host from the problem data alone."""
    g = H_mul(x) + f
    assert np.all(x >= lo - 1e-9) and np.all(x <= hi + 1e-9)
    free = (x > lo + 1e-7) & (x < hi - 1e-7)
    assert np.max(np.abs(g[free])) < tol
    assert np.all(g[x <= lo + 1e-7] > -tol) and np.all(g[x >= hi - 1e-7] < tol)


@pytest.mark.parametrize("n", [300, 777, 2100])
def test_shortest_path_blocks_and_long_rings(emu, n):
    """The scalar tridiagonal route with real blocks (n = 300, 777: 2 / 4 rows per thread, a partial last block) and on workspace vectors
    (n = 2100 > 2048: the LDS arrays do not hold the ring).  Checked by a KKT certificate computed on the host from H, f and the box
    (the dense oracle needs minutes at these sizes); n = 300 also against the dense oracle."""
    from global_racetrajectory_optimization_amd import synthetic
    ref, nv, _ = synthetic.oval_batch(1, n=n, first=7000 + n, perturb_centreline=True)
    ref, nv = ref[0], nv[0]
    w_veh = 2.0
    al, _, st, info = emu.solve_batch([dict(reftrack=ref, normvec=nv, scaling=None, kappa_bound=1.0, w_veh=w_veh)],
                                      objective=engine.OBJ_SHORTEST_PATH)
    assert st[0] == 0
    x = al[0]
    p = ref[:, :2]
    hd = 4.0 * np.sum(nv * nv, axis=1)
    hu = -2.0 * np.sum(nv * np.roll(nv, -1, axis=0), axis=1)
    f = 2.0 * np.sum(nv * (2.0 * p - np.roll(p, 1, axis=0) - np.roll(p, -1, axis=0)), axis=1)
    lo, hi = -np.maximum(ref[:, 3] - w_veh / 2, 0.001), np.maximum(ref[:, 2] - w_veh / 2, 0.001)
    _kkt_box_certificate(lambda v: hd * v + hu * np.roll(v, -1) + np.roll(hu, 1) * np.roll(v, 1), f, lo, hi, x, 1e-8 * np.max(np.abs(f)))
    assert 0 < info[0]["n_active_box"] < n
    if n == 300:
        assert np.max(np.abs(x - tph_ref.opt_shortest_path(ref, nv, w_veh))) < 1e-9


def test_rings_between_2048_and_2208_waypoints(emu):
    """Round-3 code ran rings of 2049 .. 2208 waypoints through a two-chunk LDS route whose second chunk read the first chunk's right-hand
    sides where it needed its forward sweep (status 2 / metres of error; no test covered the range).  Round 4: every ring above 2048
    waypoints takes the tridiagonal sweeps on workspace vectors.  Against CPU-B (independent assembly and solver)."""
    from global_racetrajectory_optimization_amd import synthetic
    from oracle import banded_ref
    n = 2100
    ref, nv, sc = synthetic.oval_batch(1, n=n, first=1000 + n, perturb_centreline=True)
    a_cpu, c_cpu, st_cpu, _, _ = banded_ref.solve_batch(ref, nv, sc, 0.5, 3.0)
    al, curv, st, info = emu.solve_batch([dict(reftrack=ref[0], normvec=nv[0], scaling=sc[0], kappa_bound=0.5, w_veh=3.0)])
    assert st[0] == 0 and st_cpu[0] == 0
    assert np.max(np.abs(al[0] - a_cpu[0])) < 1e-8
    assert abs(curv[0] - c_cpu[0]) < 1e-9


def test_shortest_path_golden_and_errors(emu, golden):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "shortest_path.npz"))
    g = golden["rounded_rectangle"]
    al, _, st, _ = emu.solve_batch([dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=None, kappa_bound=1.0,
                                         w_veh=float(z["w_veh"]))], objective=engine.OBJ_SHORTEST_PATH)
    assert st[0] == 0
    assert np.max(np.abs(al[0] - z["rounded_rectangle_alpha"])) < 1e-9
    # round 6: a shortest-path problem that runs out of the caller's round budget (one round here) goes on from its working set under the
    # single-pivot backup rule, which terminates on box rows -- status 0 and the golden vertex, where rounds 1-5 returned MCQ_ITER_CAP (this
    # objective has no Goldfarb-Idnani path behind it)
    went_on = 0
    for name in ("rounded_rectangle", "handling_track", "modena_2019", "berlin_2018"):
        gg = golden[name]
        al1, _, st1, info1 = emu.solve_batch([dict(reftrack=gg["reftrack"], normvec=gg["normvec"], scaling=None, kappa_bound=1.0,
                                                   w_veh=float(z["w_veh"]))], objective=engine.OBJ_SHORTEST_PATH, max_as_iter=1)
        assert st1[0] == 0 and info1[0]["gi_iters"] == 0, (name, st1[0], info1[0])
        assert np.max(np.abs(al1[0] - z[name + "_alpha"])) < 1e-8, name
        went_on += 1 if info1[0]["second_attempt"] & 4 else 0
    assert went_on >= 1            # (at least one of them needs more than the one round it was given)
    bad = g["reftrack"].copy()
    bad[3, 0] = np.nan
    _, _, st, _ = emu.solve_batch([dict(reftrack=bad, normvec=g["normvec"], scaling=None, kappa_bound=1.0, w_veh=3.4)],
                                  objective=engine.OBJ_SHORTEST_PATH)
    assert st[0] == engine.STATUS_BAD_INPUT
    with pytest.raises(engine.EngineError, match="normvec is required"):
        emu.solve_batch([dict(reftrack=g["reftrack"], normvec=None, scaling=None, kappa_bound=1.0, w_veh=3.4)],
                        objective=engine.OBJ_SHORTEST_PATH)


def test_fp32_boundary_is_exact_on_the_rounded_inputs(emu, golden):
    """mcq_solve_device_f32 (BASELINE config 5's boundary): float tracks in, float alpha out, fp64 arithmetic inside --
    the result must be the dense oracle's solution OF THE ROUNDED ROWS to one float rounding of alpha; the distance to
    the fp64-input golden alpha is the QP's sensitivity to the input rounding, far larger and not the engine's."""
    g = golden["rounded_rectangle"]
    ref32 = g["reftrack"].astype(np.float32)
    a32, curv, st, info = emu.solve_uniform_f32(ref32[None], None, None, 0.12, 3.4)
    assert a32.dtype == np.float32 and st[0] == 0
    r64 = ref32.astype(np.float64)
    _, _, A, nv_d = tph_ref.calc_splines(np.vstack((r64[:, :2], r64[0, :2])))
    a_ref, err_ref = tph_ref.opt_min_curv(r64, nv_d, A, 0.12, 3.4)
    assert np.max(np.abs(a32[0] - a_ref)) <= np.max(np.abs(a_ref)) * 2.0 ** -24 + 1e-9
    assert abs(curv[0] - err_ref) < 1e-10
    assert 1e-7 < np.max(np.abs(a32[0] - g["alpha"])) < 1e-4       # input rounding at |x|, |y| ~ 100 m
    # float normals / scalings handed over instead of derived: same solution to the input-rounding level
    b32, _, stb, _ = emu.solve_uniform_f32(ref32[None], g["normvec"][None], g["scaling"][None], 0.12, 3.4)
    assert stb[0] == 0 and np.max(np.abs(b32[0] - g["alpha"])) < 1e-4


def test_raceline_kernel_and_ragged_velocity_profiles(emu, golden):
    """The chain main_globaltraj.py runs after the QP [REF main_globaltraj.py:371-422], on the device for tracks of different
    lengths: mcq_raceline_kernel (create_raceline + calc_head_curv_an) against the host shims, then its padded kappa /
    el_lengths rows straight into the ragged velocity-profile entry, against the host chain variant by variant."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import create_raceline as cr, calc_head_curv_an as ch, \
        calc_vel_profile as cv
    names = ("rounded_rectangle", "handling_track")
    refs = [golden[k]["reftrack"] for k in names]
    nvs = [golden[k]["normvec"] for k in names]
    als = [golden[k]["alpha"] for k in names]
    out = emu.raceline_batch(refs, nvs, als, 2.0)
    assert list(out["status"]) == [0, 0]
    host = []
    for k in range(2):
        rl, _, cx, cy, inds, tv, _, _, el = cr.create_raceline(refs[k][:, :2], nvs[k], als[k], 2.0)
        psi, kap = ch.calc_head_curv_an(cx, cy, inds, tv)
        m = int(out["m"][k])
        assert m == rl.shape[0]
        dpsi = np.abs(out["psi"][k, :m] - psi)
        assert np.max(np.abs(out["xy"][k, :m] - rl)) < 1e-10
        assert np.max(np.minimum(dpsi, 2 * np.pi - dpsi)) < 1e-11
        assert np.max(np.abs(out["kappa"][k, :m] - kap)) < 1e-12
        assert np.max(np.abs(out["el_lengths"][k, :m] - el)) < 1e-10
        host.append((kap, el))
    # a raceline that needs more points than the output rows hold is reported, not truncated
    short = emu.raceline_batch(refs[:1], nvs[:1], als[:1], 2.0, mmax=50)
    assert short["status"][0] == engine.STATUS_BAD_INPUT
    # (track x vehicle) variants over the two racelines in one launch
    var = _vehicle_variants()
    track_of = np.array([0, 1, 0, 1, 1, 0], dtype=np.int32)
    pick = [var[k % 3] for k in range(6)]
    vx_d, lt_d = emu.vel_profile_batch(out["kappa"], out["el_lengths"], np.stack([v[0] for v in pick]),
                                       np.stack([v[1] for v in pick]), [v[2] for v in pick], [v[3] for v in pick],
                                       [v[4] for v in pick], dyn_model_exp=1.0, track_of=track_of, n_of_track=out["m"])
    from oracle import vel_ref
    for k, (gg, axm, drag, mass, vmax) in enumerate(pick):
        kap, el = host[track_of[k]]
        vx_o = vel_ref.calc_vel_profile(ax_max_machines=axm, kappa=kap, el_lengths=el, closed=True, drag_coeff=drag, m_veh=mass,
                                        ggv=gg, v_max=vmax, dyn_model_exp=1.0)
        assert np.max(np.abs(vx_d[k, :kap.size] - vx_o)) < 1e-8
        assert abs(lt_d[k] - vel_ref.lap_time_stable(vx_o, el)) < 1e-8


def _crossing_cases(golden):
    """Tracks whose normals do and do not cross: a reference track as is, the same with the widths blown up beyond the radius
    of its corners, a tight circle (width > radius on the inside), and a ring shorter than the horizon."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs
    g = golden["rounded_rectangle"]
    wide = g["reftrack"].copy()
    wide[:, 2:] = 60.0
    th = np.linspace(0.0, 2 * np.pi, 40, endpoint=False)
    xy = 6.0 * np.column_stack((np.cos(th), np.sin(th)))
    _, _, _, nv_c = cs.calc_splines(path=np.vstack((xy, xy[0])))
    circle = np.column_stack((xy, np.full(40, 8.0), np.full(40, 8.0)))
    return [(g["reftrack"], g["normvec"]), (wide, g["normvec"]), (circle, nv_c), (circle[:8], nv_c[:8])]


def test_normals_crossing_kernel_matches_oracle(emu, golden):
    """f-2's second half against the ORACLE's restatement of tph.check_normals_crossing (pairwise 2 x 2 solves, collinear
    normals skipped, bounds included); the host shim main_globaltraj.py calls must say the same."""
    from oracle import vel_ref
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import check_normals_crossing as cn
    cases = _crossing_cases(golden)
    got = emu.normals_crossing_batch([c[0] for c in cases], [c[1] for c in cases], horizon=10)
    want = [int(vel_ref.check_normals_crossing(t, nv, 10)) for t, nv in cases[:3]]
    assert list(got[:3]) == want and want[0] == 0 and want[1] == 1
    assert [int(cn.check_normals_crossing(t, nv, 10)) for t, nv in cases[:3]] == want
    for f in (vel_ref.check_normals_crossing, cn.check_normals_crossing):
        with pytest.raises(RuntimeError, match="too large"):
            f(cases[3][0], cases[3][1], 10)
    assert got[3] == -1
    # the bounds are part of the segments: widths shrunk until the two outermost crossing normals only just touch
    ref, nv = cases[1]
    lam = []
    n = ref.shape[0]
    for i in range(n):
        for d in range(1, 11):
            j = (i + d) % n
            M = np.column_stack((nv[i], -nv[j]))
            if abs(np.linalg.det(M)) > 1e-8:
                l = np.linalg.solve(M, ref[j, :2] - ref[i, :2])
                lam.append(max(abs(l[0]), abs(l[1])))
    w_touch = min(lam)
    for w, expect in ((w_touch * (1 + 1e-9), 1), (w_touch * (1 - 1e-9), 0)):
        trk = ref.copy()
        trk[:, 2:] = w
        assert int(vel_ref.check_normals_crossing(trk, nv, 10)) == expect
        assert int(emu.normals_crossing_batch([trk], [nv], horizon=10)[0]) == expect
        assert int(cn.check_normals_crossing(trk, nv, 10)) == expect


def test_iqp_device_resident_with_warm_started_passes(emu, golden):
    """A whole device-resident IQP run through the unchanged kernels: QP pass, glue kernel (which also carries the working set to
    the re-sampled ring), passes 2+ warm-started from it (exchange only, no interior point) -- against the golden IQP end
    state, which the dense oracle + host glue produced."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler as iq
    g = golden["rounded_rectangle"]
    st = {}
    out = iq.iqp_handler_batch([dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"])], 0.12, 3.4,
                               3.0, 3, 0.01, engine=emu, stats=st, device_resident=True, warm_start=True)
    a, r, nv = out[0]
    assert st["rounds"] == 3 and a.shape == g["iqp_alpha"].shape
    assert np.max(np.abs(a - g["iqp_alpha"])) < 1e-8
    assert np.max(np.abs(r - g["iqp_reftrack"])) < 1e-8 and np.max(np.abs(nv - g["iqp_normvec"])) < 1e-8


def test_iqp_print_debug_lines_stream_per_round(emu, golden, capsys):
    """tph.iqp_handler prints one line per iteration as it goes [REF main_globaltraj.py:270,280].  The engine runs the whole handler as
    one call; with print_debug it calls back after every QP pass (mcq_iqp_set_round_callback), so the lines appear per round -- and for
    as many rounds as there are (round 3 printed them after the run, the first 16 only).  iters_min = 20 forces 20 damped rounds."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler as iq
    g = golden["rounded_rectangle"]
    seen = []
    orig = emu.set_iqp_round_callback

    def spy(fn):                                   # records WHEN the handler's callback fires relative to the engine call
        if fn is None:
            return orig(None)
        return orig(lambda rnd, curv, live: (seen.append((rnd, float(curv[0]), int(live[0]))), fn(rnd, curv, live)))
    emu.set_iqp_round_callback = spy
    try:
        out = iq.iqp_handler_batch([dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"])], 0.12, 3.4,
                                   3.0, 20, 0.01, print_debug=True, engine=emu, device_resident=True)
    finally:
        emu.set_iqp_round_callback = orig
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("Minimum curvature IQP")]
    assert [r for r, _, _ in seen] == list(range(1, 21)) and all(l == 1 for _, _, l in seen)
    assert len(lines) == 20 and lines[0].startswith("Minimum curvature IQP: iteration 1, curv_error_max: ")
    assert lines[19].startswith("Minimum curvature IQP: iteration 20, curv_error_max: %.4frad/m" % seen[19][1])
    assert out[0][0].shape[0] == out[0][1].shape[0]
    # ... and the callback is gone afterwards: a run without print_debug prints nothing
    iq.iqp_handler_batch([dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"])], 0.12, 3.4, 3.0, 3, 0.01,
                         engine=emu)
    assert "Minimum curvature IQP" not in capsys.readouterr().out


def test_solve_host_entry_and_reopt_corridor(emu, golden):
    """mcq_solve_host (uniform batch straight from / to host arrays, the wall bench.py reports as host_to_host) returns what
    mcq_solve_batch returns, bitwise; the problems are the reference's re-optimisation consumer [REF main_globaltraj.py:337-350]
    -- widths 1.0 / 1.0, w_veh = 1.6: a +-0.2 m corridor -- and the standard configuration, against the live dense oracle."""
    g = golden["rounded_rectangle"]
    ref_c = g["reftrack"].copy()
    ref_c[:, 2:] = 1.0
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs
    A = cs.build_les_matrix(ref_c.shape[0], g["scaling"])
    a_ref, err_ref = tph_ref.opt_min_curv(ref_c, g["normvec"], A, 0.12, 1.6)
    refs = np.stack((ref_c, ref_c))
    refs[1, :, 2:] = 1.05
    al_h, curv_h, st_h, info_h = emu.solve_host(refs, np.stack((g["normvec"],) * 2), np.stack((g["scaling"],) * 2), 0.12, 1.6)
    al_b, curv_b, st_b, info_b = emu.solve_batch([dict(reftrack=refs[k], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12,
                                                       w_veh=1.6) for k in range(2)])
    assert list(st_h) == [0, 0] and list(st_b) == [0, 0]
    assert np.array_equal(al_h[0], al_b[0]) and np.array_equal(al_h[1], al_b[1]) and np.array_equal(curv_h, curv_b)
    assert np.max(np.abs(al_h[0] - a_ref)) < 1e-8 and abs(curv_h[0] - err_ref) < 1e-9
    assert np.all(np.abs(al_h[0]) <= 0.2 + 1e-12) and info_h[0].n_active_box == int(np.sum(np.abs(np.abs(a_ref) - 0.2) < 1e-9))
    # pinned host memory of the engine behaves like any other host array
    p = emu.host_array((2, ref_c.shape[0], 4))
    p[...] = refs
    al_p, _, st_p, _ = emu.solve_host(p, np.stack((g["normvec"],) * 2), np.stack((g["scaling"],) * 2), 0.12, 1.6)
    assert list(st_p) == [0, 0] and np.array_equal(al_p, al_h)


def test_curvature_rows_switch_and_warm_start_bookkeeping(emu):
    """mcq_opts.check_kappa: 0 (a zero-initialised struct) and 1 carry the curvature rows, a negative value skips them and
    REPORTS a violated row (status 6) instead of returning an infeasible alpha as OK.  mcq_opts.warm_start without
    working sets of the same batch layout at hand is ignored (cold path), never misread."""
    ref, nv, A, sc = _small_track(40, seed=5)
    a_box, _, I = tph_ref.opt_min_curv(ref, nv, A, 10.0, 2.0, return_internals=True)
    kb = 0.9 * float(np.max(np.abs(I["k_ref"] + I["E"] @ a_box)))
    a_ref, _ = tph_ref.opt_min_curv(ref, nv, A, kb, 2.0)
    p = dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=kb, w_veh=2.0)
    for ck, want in ((0, 0), (1, 0), (-1, engine.STATUS_KAPPA_ACTIVE)):
        al, _, st, _ = emu.solve_batch([p], check_kappa=ck)
        assert st[0] == want, (ck, st[0])
        if want == 0:
            assert np.max(np.abs(al[0] - a_ref)) < 1e-8
        else:
            assert np.max(np.abs(al[0] - a_box)) < 1e-8          # the box optimum, flagged
    al_w, _, st_w, info_w = emu.solve_batch([p, p], warm_start=1)    # nothing carried over for this layout: cold path
    assert list(st_w) == [0, 0] and info_w[0]["ipm_iters"] > 0 and np.max(np.abs(al_w[1] - a_ref)) < 1e-8


def test_thirteen_active_curvature_rows_with_refinement(emu, golden):
    """The curvature-row working set beyond a handful of rows (Schur complement in HBM, LU in the LDS overlay by all threads, one
    banded solve per active row + one for the correction, refinement on the KKT system): handling track, kappa_bound 0.06 ->
    13 active curvature rows next to 25 box rows, against dense Goldfarb-Idnani.  (The GPU suite runs 40 and 51 rows.)"""
    from oracle import qp_ref
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_splines as cs
    g = golden["handling_track"]
    n = g["reftrack"].shape[0]
    info = {}
    a_ref, err_ref = tph_ref.opt_min_curv(g["reftrack"], g["normvec"], cs.build_les_matrix(n, g["scaling"]), 0.06, 3.4,
                                          solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
    nk = int(np.sum(info["lagr"][2 * n:] > 0))
    al, curv, st, inf = emu.solve_batch([dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.06,
                                              w_veh=3.4)])
    assert st[0] == 0 and nk == 13 and inf[0]["n_active_kappa"] == nk
    assert np.max(np.abs(al[0] - a_ref)) < 1e-9 and abs(curv[0] - err_ref) < 1e-10
    assert abs(inf[0]["kappa_max"] - 0.06) < 1e-12 and inf[0]["refine_rounds"] >= 1


def test_poisoned_workspaces_and_lds(emu, emu_lib, golden, monkeypatch):
    """MCQ_POISON=1: every workspace / staging allocation and the solver kernel's whole LDS start out as NaN bit patterns, so anything a
    phase reads without having written it surfaces on every run instead of on the boxes whose stale memory happens to hold a NaN
    (round 2: the backward sweep behind the fused forward substitution read a chunk-ring slot only the plain forward sweep clears)."""
    monkeypatch.setenv("MCQ_POISON", "1")
    eng = engine.Engine(0, lib_path=emu_lib)
    try:
        names = ("rounded_rectangle", "handling_track")
        al, curv, st, info = eng.solve_batch([_problem(golden[k]) for k in names])
        for k, name in enumerate(names):
            assert st[k] == 0 and np.max(np.abs(al[k] - golden[name]["alpha"])) < 1e-9
        # curvature rows active (the exchange + refinement path) and a warm-started IQP under the same poison
        g = golden["rounded_rectangle"]
        al2, curv2, st2, _ = eng.solve_batch([dict(_problem(g), kappa_bound=0.10)])
        al3, curv3, st3, _ = emu.solve_batch([dict(_problem(g), kappa_bound=0.10)])
        assert st2[0] == st3[0] and np.array_equal(al2[0], al3[0]) and np.array_equal(curv2, curv3)     # bitwise: nothing stale was read
        from global_racetrajectory_optimization_amd.trajectory_planning_helpers import iqp_handler as iq
        out = iq.iqp_handler_batch([dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"])], 0.12, 3.4,
                                   3.0, 3, 0.01, engine=eng, device_resident=True, warm_start=True)
        assert np.max(np.abs(out[0][0] - g["iqp_alpha"])) < 1e-8
    finally:
        eng.close()


def test_iqp_batch_into_caller_kept_buffers(emu, golden):
    """Engine.iqp_batch(out=...): the end states land in arrays the caller keeps across calls (page-locked ones from host_array in
    production); same results as the call that allocates its own, wrong shapes are refused."""
    g = golden["rounded_rectangle"]
    trk = [dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"])] * 2
    a = emu.iqp_batch(trk, 0.12, 3.4, 3.0, 3, 0.01)
    nmax = a["stats"]["nmax"]
    buf = dict(alpha=emu.host_array((2, nmax)), reftrack=emu.host_array((2, nmax, 4)), normvectors=emu.host_array((2, nmax, 2)))
    b = emu.iqp_batch(trk, 0.12, 3.4, 3.0, 3, 0.01, nmax=nmax, out=buf)
    for k in range(2):
        assert np.array_equal(a["alpha"][k], b["alpha"][k]) and np.array_equal(a["reftrack"][k], b["reftrack"][k])
        assert np.shares_memory(b["alpha"][k], buf["alpha"])
        assert np.max(np.abs(b["alpha"][k] - g["iqp_alpha"])) < 1e-8
    with pytest.raises(ValueError):
        emu.iqp_batch(trk, 0.12, 3.4, 3.0, 3, 0.01, nmax=nmax, out=dict(alpha=np.zeros((2, nmax + 1))))
    # a uniform batch as ONE dict of stacked arrays (no per-track Python work): the same end states
    c = emu.iqp_batch(dict(reftrack=np.stack([g["reftrack"]] * 2), normvectors=np.stack([g["normvec"]] * 2), scaling=np.stack([g["scaling"]] * 2)),
                      0.12, 3.4, 3.0, 3, 0.01)
    for k in range(2):
        assert np.array_equal(a["alpha"][k], c["alpha"][k]) and np.array_equal(a["normvectors"][k], c["normvectors"][k])
    with pytest.raises(ValueError):
        emu.iqp_batch(dict(reftrack=np.zeros((2, 50, 4)), normvectors=np.zeros((2, 49, 2))), 0.12, 3.4, 3.0)


def test_iqp_rounds_in_one_launch_equal_the_round_by_round_loop(emu, golden, monkeypatch):
    """mcq_iqp_device runs the first iters_min rounds as ONE launch in which every workgroup takes its track through the rounds on its own
    (mcq_iqp_rounds_kernel: the bodies of the solver, bookkeeping and glue kernels between workgroup barriers, the same arrays; round 5).
    Six tracks of two shapes against the one-launch-per-round loop ($MCQ_IQP_FUSED=0): end states bitwise, round counts and statuses equal --
    tracks that need a fourth round (the loop takes over after round iters_min), one whose QP is infeasible from the start (it stops in
    round 1: neither of its ring buffers may be solved again), cold passes (warm_start=-1) as well."""
    trk = []
    for name, dw in (("rounded_rectangle", 0.0), ("handling_track", 0.0), ("rounded_rectangle", 0.4), ("handling_track", -0.2),
                     ("rounded_rectangle", -0.3), ("handling_track", 0.3)):
        g = golden[name]
        r = g["reftrack"].copy()
        r[:, 2:] += dw
        trk.append(dict(reftrack=r, normvectors=g["normvec"], scaling=g["scaling"]))
    trk[4]["reftrack"][:, 2:] = 1.0          # narrower than the vehicle: status 1 in the first pass
    for kw in (dict(), dict(warm_start=-1)):
        res = {}
        for mode, fused in (("loop", "0"), ("one launch", "1")):
            monkeypatch.setenv("MCQ_IQP_FUSED", fused)
            res[mode] = emu.iqp_batch(trk if not kw else trk[:2], 0.12, 3.4, 3.0, 3, 3e-3, **kw)   # (curv_error_allowed 0.003: a fourth round)
        a, b = res["loop"], res["one launch"]
        ntr = len(a["status"])
        if not kw:
            assert a["status"][4] == engine.STATUS_INFEASIBLE and a["rounds"][4] == 1
        assert max(a["rounds"]) > 3, list(a["rounds"])
        assert list(a["status"]) == list(b["status"]) and list(a["rounds"]) == list(b["rounds"]) and list(a["n"]) == list(b["n"])
        for k in range(ntr):
            for key in ("alpha", "reftrack", "normvectors"):
                assert np.array_equal(a[key][k], b[key][k]), (kw, k, key)
            assert a["curv_err"][k] == b["curv_err"][k] and np.array_equal(a["curv_trace"][k], b["curv_trace"][k]), (kw, k)
        assert a["stats"]["qp_solves"] == b["stats"]["qp_solves"] == int(np.sum(a["rounds"]))


def test_host_batches_packed_by_several_threads(emu, golden, monkeypatch):
    """mcq_solve_batch / mcq_iqp_batch pack large batches into the pinned staging in chunks of tracks on several host threads, each chunk's
    uploads queued as it is ready ($MCQ_PACK_THREADS forces it for a small batch): a ragged batch -- scalings given for some tracks only,
    more threads than some chunks have tracks -- returns exactly what the single loop returns."""
    probs = []
    for k, name in enumerate(("rounded_rectangle", "handling_track", "rounded_rectangle", "handling_track", "rounded_rectangle", "rounded_rectangle", "handling_track")):
        g = golden[name]
        r = g["reftrack"].copy()
        r[:, 2:] += 0.05 * k
        probs.append(dict(reftrack=r, normvec=g["normvec"], scaling=g["scaling"] if k % 3 != 1 else None, kappa_bound=0.12, w_veh=3.4 - 0.1 * (k % 2)))
    res = {}
    for th in ("1", "3"):
        monkeypatch.setenv("MCQ_PACK_THREADS", th)
        res[th] = emu.solve_batch(probs)
    for k in range(len(probs)):
        assert np.array_equal(res["1"][0][k], res["3"][0][k]), k
    assert np.array_equal(res["1"][1], res["3"][1]) and list(res["1"][2]) == list(res["3"][2])
    assert res["1"][2][0] == 0 and np.max(np.abs(res["1"][0][0] - golden["rounded_rectangle"]["alpha"])) < 1e-8


def test_fp32_increment_rows_keep_the_accuracy(emu, golden):
    """MCQ_F32_INCREMENTS (round 3): float rows [x_{i+1} - x_i, y_{i+1} - y_i, w_r, w_l] + an fp64 origin per track.  (i) the engine
    solves EXACTLY the QP of the rows it rebuilds (fp64 running sum, closure defect spread over the ring): against the dense oracle
    on engine.increments_to_rows(...) to one float rounding of alpha; (ii) that QP is far closer to the fp64-input one than with
    absolute float coordinates -- the stated reason for the layout; (iii) layout 0 through the same entry equals the old float entry;
    (iv) the origin does not enter alpha."""
    g = golden["rounded_rectangle"]
    ref = g["reftrack"].copy()
    ref[:, :2] += np.array([1500.0, -900.0])          # a track far from the origin: absolute float coordinates lose 1.2e-4 m
    rows32, org = engine.rows_to_increments(ref[None])
    a_inc, curv, st, info = emu.solve_batch_f32(rows32, org, 0.12, 3.4, layout=engine.F32_INCREMENTS)
    assert a_inc.dtype == np.float32 and st[0] == 0
    r64 = engine.increments_to_rows(rows32, org)[0]
    assert np.max(np.abs(r64[:, :2] - ref[:, :2])) < 1e-5
    _, _, A, nv_d = tph_ref.calc_splines(np.vstack((r64[:, :2], r64[0, :2])))
    a_ref, err_ref = tph_ref.opt_min_curv(r64, nv_d, A, 0.12, 3.4)
    assert np.max(np.abs(a_inc[0] - a_ref)) <= np.max(np.abs(a_ref)) * 2.0 ** -24 + 1e-9
    assert abs(curv[0] - err_ref) < 1e-10
    d_inc = float(np.max(np.abs(a_inc[0] - g["alpha"])))
    a_abs, _, st_a, _ = emu.solve_batch_f32(ref[None].astype(np.float32), None, 0.12, 3.4, layout=engine.F32_ABSOLUTE)
    d_abs = float(np.max(np.abs(a_abs[0] - g["alpha"])))
    assert st_a[0] == 0 and d_inc < 2e-6 and d_abs > 20 * d_inc, (d_inc, d_abs)
    a_old, _, st_o, _ = emu.solve_uniform_f32(ref[None].astype(np.float32), None, None, 0.12, 3.4)
    assert st_o[0] == 0 and np.array_equal(a_old, a_abs)
    a_no_org, _, _, _ = emu.solve_batch_f32(rows32, None, 0.12, 3.4, layout=engine.F32_INCREMENTS)
    assert np.max(np.abs(a_no_org[0].astype(np.float64) - a_inc[0])) < 1e-6


def test_solve_host_in_slices_is_bitwise_the_one_launch(emu, monkeypatch):
    """Round 6: mcq_solve_host takes a batch of 512 or more (here: $MCQ_HOST_SLICE_MIN = 8) in SLICES (two by default, four here) -- upload k + 1 / kernel k / download k - 1 overlapped, the kernels of
    consecutive slices on the handle's two compute streams, all on disjoint rows of one workspace (McqBatch.pb0) -- and must return what the one
    launch ($MCQ_HOST_ONE_LAUNCH=1) returns, bit for bit, info records included; a narrow corridor in slice 2 keeps its status."""
    monkeypatch.setenv("MCQ_HOST_SLICE_MIN", "4")
    monkeypatch.setenv("MCQ_HOST_SLICES", "4")          # (the default is two; four exercises a stream's second slice)
    n, bsz = 24, 7                          # (not a multiple of four: the slices differ in size)
    base = [_small_track(n, seed=300 + k) for k in range(5)]
    rng = np.random.default_rng(9)
    refs = np.stack([base[k % 5][0] for k in range(bsz)])
    refs[:, :, 2:] += rng.uniform(0.0, 0.8, size=(bsz, n, 2))
    refs[5, 3, 2:] = 0.5                     # w_r + w_l < w_veh: MCQ_INFEASIBLE for this one
    nvs = np.stack([base[k % 5][1] for k in range(bsz)])
    scs = np.stack([base[k % 5][3] for k in range(bsz)])
    al, cu, st, info = emu.solve_host(refs, nvs, scs, 0.5, 2.0)
    monkeypatch.setenv("MCQ_HOST_ONE_LAUNCH", "1")
    al1, cu1, st1, info1 = emu.solve_host(refs, nvs, scs, 0.5, 2.0)
    assert st[5] == engine.STATUS_INFEASIBLE and np.count_nonzero(st) == 1 and np.array_equal(st, st1)
    assert np.array_equal(al, al1) and np.array_equal(cu, cu1)
    assert [i.ipm_iters for i in info] == [i.ipm_iters for i in info1] and [i.as_iters for i in info] == [i.as_iters for i in info1]
    # ... and what the ragged host-buffer entry returns for three of them
    probs = [dict(reftrack=refs[k], normvec=nvs[k], scaling=scs[k], kappa_bound=0.5, w_veh=2.0) for k in (0, 4, 6)]
    al2, _, st2, _ = emu.solve_batch(probs)
    assert all(np.array_equal(al2[j], al[k]) for j, k in enumerate((0, 4, 6)))
    # the ragged host-buffer entry (mcq_solve_batch: what the drop-in's opt_min_curv_batch calls) slices the same way: seven rings of three
    # different sizes, per-problem vehicle widths, one of them infeasible -- sliced and in one launch, bit for bit
    rag = []
    for k in range(7):
        r_, v_, _, s_ = _small_track(20 + 3 * (k % 3), seed=400 + k)
        rag.append(dict(reftrack=r_, normvec=v_, scaling=s_, kappa_bound=0.5, w_veh=2.0 + 0.05 * k))
    rag[3]["reftrack"] = rag[3]["reftrack"].copy()
    rag[3]["reftrack"][2, 2:] = 0.4
    a_one, c_one, s_one, i_one = emu.solve_batch(rag)           # (MCQ_HOST_ONE_LAUNCH is still set)
    monkeypatch.delenv("MCQ_HOST_ONE_LAUNCH")
    a_sl, c_sl, s_sl, i_sl = emu.solve_batch(rag)
    assert list(s_sl) == list(s_one) and s_sl[3] == engine.STATUS_INFEASIBLE and np.count_nonzero(s_sl) == 1
    assert all(np.array_equal(x, y) for x, y in zip(a_sl, a_one)) and np.array_equal(c_sl, c_one)
    assert [i["ipm_iters"] for i in i_sl] == [i["ipm_iters"] for i in i_one]


def test_uniform_pinned_batches_skip_the_packing_pass(emu, golden, monkeypatch):
    """Round 6: a uniform batch whose arrays lie in page-locked memory as one contiguous block each (rows of engine.host_array blocks; what a batch
    service keeps between calls) is uploaded straight from there -- one strided copy per array, rows of n waypoints into rows of nmax -- instead
    of being packed into the staging first.  mcq_solve_batch (one launch and in slices) and mcq_iqp_batch (whose device rows are LONGER than the
    caller's: nmax > n) must return what the packing pass ($MCQ_PACK_ALWAYS=1) returns, bit for bit; pageable copies of the same arrays take the
    packing pass by themselves."""
    g = golden["rounded_rectangle"]
    t_ref, t_nv, _, t_sc = _small_track(30, seed=77)
    n, bsz = 30, 5
    p_ref, p_nv, p_sc = emu.host_array((bsz, n, 4)), emu.host_array((bsz, n, 2)), emu.host_array((bsz, n))
    rng = np.random.default_rng(2)
    for k in range(bsz):
        p_ref[k] = t_ref
        p_ref[k, :, 2:] += rng.uniform(0.0, 0.6, size=(n, 2))
        p_nv[k], p_sc[k] = t_nv, t_sc
    probs = [dict(reftrack=p_ref[k], normvec=p_nv[k], scaling=p_sc[k], kappa_bound=0.5, w_veh=2.0) for k in range(bsz)]
    pageable = [dict(reftrack=p_ref[k].copy(), normvec=p_nv[k].copy(), scaling=p_sc[k].copy(), kappa_bound=0.5, w_veh=2.0) for k in range(bsz)]
    a_dir, c_dir, s_dir, _ = emu.solve_batch(probs)
    assert emu.last_upload_was_direct()
    a_pg, c_pg, s_pg, _ = emu.solve_batch(pageable)
    assert not emu.last_upload_was_direct()
    monkeypatch.setenv("MCQ_HOST_SLICE_MIN", "4")
    a_sl, c_sl, s_sl, _ = emu.solve_batch(probs)                   # direct uploads per slice
    assert emu.last_upload_was_direct()
    monkeypatch.setenv("MCQ_PACK_ALWAYS", "1")
    a_pk, c_pk, s_pk, _ = emu.solve_batch(probs)
    assert not emu.last_upload_was_direct()
    assert np.all(s_dir == 0)
    for other_a, other_c in ((a_pg, c_pg), (a_sl, c_sl), (a_pk, c_pk)):
        assert all(np.array_equal(x, y) for x, y in zip(a_dir, other_a)) and np.array_equal(c_dir, other_c)
    assert not np.array_equal(a_dir[0], a_dir[1])                    # (the widths differ: the rows did not all come from track 0)
    # the IQP call: stacked pinned arrays (two copies of the reference's smallest track), device rows of nmax > n waypoints
    n = g["reftrack"].shape[0]
    q_ref, q_nv, q_sc = emu.host_array((2, n, 4)), emu.host_array((2, n, 2)), emu.host_array((2, n))
    for k in range(2):
        q_ref[k], q_nv[k], q_sc[k] = g["reftrack"], g["normvec"], g["scaling"]
    q_ref[1, :, 2:] += 0.2
    trk = dict(reftrack=q_ref, normvectors=q_nv, scaling=q_sc)
    iq_pk = emu.iqp_batch(trk, 0.12, 3.4, 3.0, iters_min=2, curv_error_allowed=1.0)          # (two rounds: the upload is what is compared)
    assert not emu.last_upload_was_direct()
    monkeypatch.delenv("MCQ_PACK_ALWAYS")
    iq_dir = emu.iqp_batch(trk, 0.12, 3.4, 3.0, iters_min=2, curv_error_allowed=1.0)
    assert emu.last_upload_was_direct()
    assert iq_dir["stats"]["nmax"] > n and np.all(iq_dir["status"] == 0)
    assert all(np.array_equal(x, y) for x, y in zip(iq_dir["alpha"], iq_pk["alpha"]))
    assert all(np.array_equal(x, y) for x, y in zip(iq_dir["reftrack"], iq_pk["reftrack"]))


def test_solve_host_pipelined_equals_solve_host(emu, golden):
    """mcq_solve_host_pipelined (uploads / kernels / downloads of consecutive batches overlapped on three streams, two staging
    slots): every step's results are bitwise those of the blocking entry on the same buffers -- five steps, so both slots are
    reused, with different rows per step and the normals derived on the device in one of them."""
    g = golden["rounded_rectangle"]
    n = g["reftrack"].shape[0]
    refs, nvs, scs, outs = [], [], [], []
    for k in range(5):
        r = np.stack((g["reftrack"], g["reftrack"]))
        r[0, :, 2:] += 0.05 * k
        r[1, :, 2:] += 0.02 * (k + 1)
        refs.append(r)
        nvs.append(None if k == 2 else np.stack((g["normvec"],) * 2))
        scs.append(None if k == 2 else np.stack((g["scaling"],) * 2))
        outs.append(np.full((2, n), np.nan))
    curv, st = emu.solve_host_pipelined(refs, nvs, scs, 0.12, 3.4, outs)
    assert np.all(st == 0)
    for k in range(5):
        al, cu, s1, _ = emu.solve_host(refs[k], nvs[k], scs[k], 0.12, 3.4)
        assert np.array_equal(al, outs[k]) and np.array_equal(cu, curv[k]) and list(s1) == [0, 0], k
    with pytest.raises(ValueError):
        emu.solve_host(refs[0], nvs[0], scs[0], 0.12, 3.4, alpha_out=np.zeros((2, n), dtype=np.float32))


def test_resident_stream_entry_matches_single_launches(emu, golden):
    """mcq_solve_device_stream on the interpreter (its second stream and workspace are ordinary memory there): three resident batches, bitwise the
    launch-by-launch results."""
    g = golden["rounded_rectangle"]
    n = g["reftrack"].shape[0]
    ptr, want = [], []
    for k in range(3):
        ref = np.stack((g["reftrack"], g["reftrack"]))
        ref[:, :, 2:] += 0.04 * k
        nv, sc = np.stack((g["normvec"],) * 2), np.stack((g["scaling"],) * 2)
        d = [emu.alloc(a.nbytes) for a in (ref, nv, sc)] + [emu.alloc(8 * 2 * n), emu.alloc(16), emu.alloc(8)]
        for q, a in zip(d, (ref, nv, sc)):
            emu.upload(q, a)
        emu.solve_device(2, n, d[0], d[1], d[2], 0.12, 3.4, d[3], d[4], d[5])
        want.append(emu.download(d[3], (2, n), np.float64))
        emu.upload(d[3], np.zeros((2, n)))
        ptr.append(d)
    emu.solve_device_stream(2, n, [d[0] for d in ptr], [d[1] for d in ptr], [d[2] for d in ptr], 0.12, 3.4, [d[3] for d in ptr], [d[4] for d in ptr],
                            [d[5] for d in ptr])
    emu.sync()
    for k in range(3):
        assert np.array_equal(emu.download(ptr[k][3], (2, n), np.float64), want[k])
        assert list(emu.download(ptr[k][5], (2,), np.int32)) == [0, 0]


def _stadium(n, ls=120.0, r=40.0):
    """Two straights and two semicircles, n points equidistant in arclength (counter-clockwise)."""
    per = 2 * ls + 2 * np.pi * r
    xy = np.zeros((n, 2))
    for k, sk in enumerate(np.linspace(0.0, per, n, endpoint=False)):
        if sk < ls:
            xy[k] = (sk - ls / 2, -r)
        elif sk < ls + np.pi * r:
            th = (sk - ls) / r - np.pi / 2
            xy[k] = (ls / 2 + r * np.cos(th), r * np.sin(th))
        elif sk < 2 * ls + np.pi * r:
            xy[k] = (ls / 2 - (sk - ls - np.pi * r), r)
        else:
            th = (sk - 2 * ls - np.pi * r) / r + np.pi / 2
            xy[k] = (-ls / 2 + r * np.cos(th), r * np.sin(th))
    return xy


def stadium_problem(n=360, kappa_bound=0.0223):
    """A case quadprog solves and rounds 1-2 of the engine did not: on the two long arcs of a stadium the
    curvature of the optimal line sits on a plateau, and a bound just below it puts MORE curvature rows into the working set than
    the LDS-resident Schur path holds (MCQ_KMAX = 120): 134 at n = 360."""
    xy = _stadium(n)
    _, _, A, nv = tph_ref.calc_splines(np.vstack((xy, xy[0])))
    ref = np.column_stack((xy, np.full((n, 2), 4.0)))
    idx = np.arange(n - 1)
    sc = np.empty(n)
    sc[:-1] = -A[4 * idx + 2, 4 * idx + 5]
    sc[-1] = A[4 * n - 2, 1]
    return ref, nv, A, sc, kappa_bound


def test_more_curvature_rows_than_the_lds_path_holds(emu):
    """134 active curvature rows (> MCQ_KMAX): the problem claims an overflow slot of the handle -- Schur matrix and its pivoted LU in
    HBM -- and returns the dense Goldfarb-Idnani vertex with status 0 (it was MCQ_KAPPA_ACTIVE until round 3; before that the
    curvature-row interior point gave up with a rounding-induced non-positive pivot near the end of its path)."""
    from oracle import qp_ref
    ref, nv, A, sc, kb = stadium_problem()
    n = ref.shape[0]
    info = {}
    a_ref, err_ref = tph_ref.opt_min_curv(ref, nv, A, kb, 2.0, solver=lambda H, f, G, h: qp_ref.solve_qp_gi(H, f, G, h, info))
    nk = int(np.sum(info["lagr"][2 * n:] > 0))
    assert nk > 120
    al, curv, st, inf = emu.solve_batch([dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=kb, w_veh=2.0)])
    assert st[0] == 0
    assert inf[0]["n_active_kappa"] == nk
    assert np.max(np.abs(al[0] - a_ref)) < 1e-7
    assert abs(curv[0] - err_ref) < 1e-9
    assert abs(inf[0]["kappa_max"] - kb) < 1e-9


def test_iqp_ring_overflow_has_its_own_status(emu, golden):
    """a re-sampled raceline that outgrows the caller's buffers is MCQ_RING_OVERFLOW (7), not the MCQ_BAD_INPUT a
    non-finite input row gets -- and the drop-in raises the message that names the buffers only for the former."""
    from global_racetrajectory_optimization_amd import trajectory_planning_helpers as tph
    g = golden["rounded_rectangle"]
    n = g["reftrack"].shape[0]
    trk = dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"])
    out = emu.iqp_batch([trk], 0.12, 3.4, 1.5, iters_min=3, curv_error_allowed=0.01, nmax=n + 8)       # step 1.5 m: ~2 n points
    assert out["status"][0] == engine.STATUS_RING_OVERFLOW and out["rounds"][0] == 1
    bad = dict(trk, reftrack=g["reftrack"].copy())
    bad["reftrack"][3, 0] = np.nan
    out2 = emu.iqp_batch([bad], 0.12, 3.4, 3.0, iters_min=3, curv_error_allowed=0.01)
    assert out2["status"][0] == engine.STATUS_BAD_INPUT
    with pytest.raises(RuntimeError, match="non-finite input"):
        tph.iqp_handler.iqp_handler_batch([bad], 0.12, 3.4, 3.0, 3, 0.01, engine=emu)


@pytest.mark.parametrize("n,fused,pinned", [(7, 0, 0.0), (40, 1, 0.0), (105, 1, 0.0), (333, 0, 0.0), (333, 1, 0.0), (501, 1, 0.0), (333, 0, 0.3), (333, 1, 0.9)])
def test_saddle_point_elimination_in_isolation(n, fused, pinned, tmp_path):
    """The solver's linear algebra (csrc/mcq_kkt.inc) on its own: scripts/kkt_check.hip -- the same diagnostic that runs on the GPU box --
    compiled against the SIMT interpreter.  One synthetic ring, sigma over 18 decades: every repetition must reproduce the first
    solution bit for bit, and the solution must satisfy the DENSE reduced system (sig + E'E) x = r, E built from the dense inverse of the
    spline system, to a backward error of 1e-13 (measured: 3e-16 ... 2e-15) -- with the right-hand side riding through the factorisation
    (fused) and with the solve's own forward / backward chains; one segment (n < 48), a few, and all sixteen; with 30 % / 90 % of the
    waypoints pinned (rows and columns of the reduced system replaced by identity: the active-set phase's systems)."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "kc_emu")
    subprocess.run(["g++", "-O2", "-std=c++17", "-x", "c++", "-I", os.path.join(root, "tests", "emu", "include"), "-o", exe,
                    os.path.join(root, "scripts", "kkt_check.hip"), "-Wno-unused-result", "-Wno-attributes"], check=True)
    out = subprocess.run([exe, str(n), "2", "2", "12", "1", str(fused), str(pinned)], check=True, capture_output=True, text=True).stdout
    m = re.search(r"factor status (\d+), entries differing from the first solution (\d+) .* NaNs (\d+)", out)
    assert m and m.group(1) == "0" and m.group(2) == "0" and m.group(3) == "0", out
    berr = float(re.search(r"backward error .*: ([0-9.e+-]+)", out).group(1))
    assert berr < 1e-13, out


_REVERSE_SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import load_golden
from global_racetrajectory_optimization_amd import engine
eng = engine.Engine(0, lib_path=sys.argv[2])
g, h = load_golden("rounded_rectangle"), load_golden("handling_track")
out = {}
probs = [dict(reftrack=g["reftrack"], normvec=g["normvec"], scaling=g["scaling"], kappa_bound=0.12, w_veh=3.4),
         dict(reftrack=h["reftrack"], normvec=h["normvec"], scaling=h["scaling"], kappa_bound=0.055, w_veh=2.0)]      # curvature rows active
for alg in (engine.ALG_DEFAULT, engine.ALG_GI):
    al, curv, st, info = eng.solve_batch(probs, algorithm=alg)
    for k in range(2):
        out["alpha_%d_%d" % (alg, k)] = al[k]
    out["curv_%d" % alg], out["status_%d" % alg] = curv, st
r = eng.iqp_batch([dict(reftrack=g["reftrack"].copy(), normvectors=g["normvec"], scaling=g["scaling"]),
                   dict(reftrack=h["reftrack"].copy(), normvectors=h["normvec"], scaling=h["scaling"])], 0.12, 3.4, 3.0, 3, 0.01)
for k in range(2):
    out["iqp_alpha_%d" % k], out["iqp_ref_%d" % k] = r["alpha"][k], r["reftrack"][k]
out["iqp_rounds"] = r["rounds"]
np.savez(sys.argv[3], **out)
'''


def test_reversed_work_item_order_changes_nothing(emu_lib, tmp_path):
    """The interpreter runs the work-items of a workgroup in their natural order between barriers; with HIPEMU_REVERSE=1 the LAST wave runs every
    phase first.  A kernel without a missing barrier cannot tell the difference: a box problem, one with active curvature rows, both through
    the default path and the Goldfarb-Idnani path, and iqp_handler's rounds in one launch (mcq_iqp_rounds_kernel) -- bitwise the same in
    both orders (each order in a process of its own: the variable is read once)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rev.py"
    script.write_text(_REVERSE_SCRIPT)
    res = {}
    for rev in ("0", "1"):
        env = dict(os.environ, HIPEMU_REVERSE=rev, OMP_NUM_THREADS="1")
        env.pop("MCQ_LIB", None)
        p = subprocess.run([sys.executable, str(script), root, emu_lib, str(tmp_path / ("out%s.npz" % rev))], env=env, capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[rev] = np.load(tmp_path / ("out%s.npz" % rev))
    assert sorted(res["0"].files) == sorted(res["1"].files)
    for key in res["0"].files:
        assert np.array_equal(res["0"][key], res["1"][key]), key
    assert list(res["0"]["status_0"]) == [0, 0] and list(res["0"]["iqp_rounds"]) == [3, 4]
