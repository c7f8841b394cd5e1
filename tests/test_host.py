"""CPU tests of the host side: the tph shim helpers against the dense oracle, the C-ABI library exports, error paths."""
import ctypes
import os
import re

import numpy as np
import pytest

from global_racetrajectory_optimization_amd import engine, synthetic
from global_racetrajectory_optimization_amd import trajectory_planning_helpers as tph
from oracle import tph_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_calc_splines_tridiagonal_vs_dense(golden):
    for name in ("rounded_rectangle", "handling_track"):
        ref = golden[name]["reftrack"]
        path_cl = np.vstack((ref[:, :2], ref[0, :2]))
        for dist in (True, False):
            cx, cy, A, nv = tph.calc_splines.calc_splines(path_cl, use_dist_scaling=dist)
            cx2, cy2, A2, nv2 = tph_ref.calc_splines(path_cl, use_dist_scaling=dist)
            assert np.max(np.abs(cx - cx2)) < 1e-11 and np.max(np.abs(cy - cy2)) < 1e-11
            assert np.max(np.abs(A - A2)) < 1e-14
            assert np.max(np.abs(nv - nv2)) < 1e-13
            s = tph.calc_splines.scalings_from_les_matrix(A)
            assert np.max(np.abs(s - tph.calc_splines.spline_scalings(path_cl, None, dist))) < 1e-15


def test_raceline_glue_vs_oracle(golden):
    g = golden["handling_track"]
    ref, nv, alpha = g["reftrack"], g["normvec"], g["alpha"] / 3.0
    out = tph.create_raceline.create_raceline(ref[:, :2], nv, alpha, 3.0)
    out2 = tph_ref.create_raceline(ref[:, :2], nv, alpha, 3.0)
    for a, b in zip(out, out2):
        assert np.asarray(a).shape == np.asarray(b).shape
        assert np.max(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float))) < 1e-9
    w = tph.interp_track_widths.interp_track_widths(ref[:, 2:], out[4], out[5])
    w2 = tph_ref.interp_track_widths(ref[:, 2:], out2[4], out2[5])
    assert np.max(np.abs(w - w2)) < 1e-12


def test_library_exports_every_declared_symbol():
    """libmcq.so (built by hipcc for gfx950, no GPU needed to load it) exports what include/mcq.h declares."""
    hdr = open(os.path.join(ROOT, "include", "mcq.h")).read()
    declared = set(re.findall(r"\b(mcq_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mcq_handle"}
    assert set(engine.EXPORTED_SYMBOLS) == declared
    lib_path = engine.DEFAULT_LIB
    if not os.path.exists(lib_path):
        import subprocess
        subprocess.run([os.path.join(ROOT, "global_racetrajectory_optimization_amd", "csrc", "build.sh")], check=True)
    lib = ctypes.CDLL(lib_path)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_engine_fails_loudly_without_library(tmp_path):
    with pytest.raises(engine.EngineError, match="not found"):
        engine.load_library(str(tmp_path / "nope.so"))


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "global_racetrajectory_optimization_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_synthetic_oval_generator_is_deterministic():
    ref, nv, sc = synthetic.oval_batch(2, n=400)
    ref2, _, _ = synthetic.oval_batch(2, n=400)
    assert np.array_equal(ref, ref2)
    assert ref.shape == (2, 400, 4) and nv.shape == (2, 400, 2) and sc.shape == (2, 400)
    seg = np.hypot(*np.diff(np.vstack((ref[0, :, :2], ref[0, :1, :2])), axis=0).T)
    assert abs(seg.mean() - 6000.0 / 400) < 0.2 and seg.std() / seg.mean() < 0.01
    assert np.all(ref[:, :, 2:] > 3.4) and np.all(ref[:, :, 2:] < 6.6)
    assert np.max(np.abs(np.sum(nv ** 2, axis=2) - 1.0)) < 1e-12
    assert not np.array_equal(ref[0, :, 2], ref[1, :, 2])


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/mcq.h is a C ABI: it must compile as C99 with no C++ or torch types in sight, and a C translation unit that
    references every declared entry point must link against libmcq.so (no compute: nothing is run)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    hdr = open(os.path.join(ROOT, "include", "mcq.h")).read()
    declared = sorted(set(re.findall(r"\b(mcq_[a-z0-9_]+)\s*\(", hdr)) - {"mcq_handle"})
    src = tmp_path / "abi.c"
    src.write_text('#include "mcq.h"\n#include <stdio.h>\ntypedef void (*fn_t)(void);\nint main(void) {\n    fn_t p[] = {%s};\n'
                   '    printf("%%d\\n", (int)(sizeof(p) / sizeof(p[0])));\n    return p[0] == 0;\n}\n'
                   % ", ".join("(fn_t)%s" % s for s in declared))
    lib_dir = os.path.dirname(engine.DEFAULT_LIB)
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                    "-o", str(exe), "-L", lib_dir, "-l:libmcq.so", "-Wl,-rpath," + lib_dir, "-Wl,--allow-shlib-undefined"], check=True)
    assert exe.exists()


def test_les_matrix_structure_guard(golden):
    """opt_min_curv reads only the N scalings out of the dense matrix A; a matrix that is not calc_splines' closed-spline
    system must be rejected (RuntimeError), not silently misread."""
    cs = tph.calc_splines
    ref = golden["rounded_rectangle"]["reftrack"]
    path_cl = np.vstack((ref[:, :2], ref[0, :2]))
    _, _, A, nv = cs.calc_splines(path_cl)
    n = ref.shape[0]
    assert np.array_equal(cs.scalings_from_les_matrix(A), cs.spline_scalings(path_cl))
    assert np.array_equal(cs.scalings_from_les_matrix(tph_ref.calc_splines(path_cl)[2]), cs.spline_scalings(path_cl))
    for r, c, dv in ((7, 9, 0.5), (4 * n - 1, 0, 1e-3), (2, 2, -2.0), (0, 0, 0.25), (6, 9, 5.0), (40, 200, 1e-9), (7, 10, 1e-6)):
        B = A.copy()
        B[r, c] += dv
        with pytest.raises(RuntimeError, match="structure of calc_splines"):
            cs.scalings_from_les_matrix(B)
    with pytest.raises(RuntimeError, match="structure of calc_splines"):
        cs.scalings_from_les_matrix(np.eye(4 * n))
    with pytest.raises(RuntimeError, match="structure of calc_splines"):
        cs.scalings_from_les_matrix(A[::-1].copy())
    # the drop-in entry points run the guard before anything reaches the engine
    with pytest.raises(RuntimeError, match="structure of calc_splines"):
        tph.opt_min_curv.opt_min_curv(ref, nv, np.eye(4 * n), 0.12, 3.4)
    # round 6: the drop-in's guard is the C ABI's mcq_les_scalings (one threaded pass over the dense matrix; host code of libmcq.so, no GPU) --
    # the same scalings bit for bit and the same verdicts as the numpy statement above, single- and multi-threaded (Berlin: 3104 x 3104)
    from global_racetrajectory_optimization_amd import engine
    assert np.array_equal(engine.les_scalings(A), cs.scalings_from_les_matrix(A))
    assert np.array_equal(engine.les_scalings(np.asfortranarray(A)), cs.scalings_from_les_matrix(A))       # (not C-contiguous: the numpy statement serves)
    for r, c, dv in ((7, 9, 0.5), (4 * n - 1, 0, 1e-3), (2, 2, -2.0), (0, 0, 0.25), (6, 9, 5.0), (40, 200, 1e-9), (7, 10, 1e-6), (4 * n - 2, 1, -50.0)):
        B = A.copy()
        B[r, c] += dv
        with pytest.raises(RuntimeError, match="structure of calc_splines"):
            engine.les_scalings(B)
    for bad in (np.eye(4 * n), A[::-1].copy()):
        with pytest.raises(RuntimeError, match="structure of calc_splines"):
            engine.les_scalings(bad)
    refb = golden["berlin_2018"]["reftrack"]
    Ab = cs.calc_splines(np.vstack((refb[:, :2], refb[0, :2])))[2]
    assert np.array_equal(engine.les_scalings(Ab), cs.scalings_from_les_matrix(Ab)) and np.array_equal(engine.les_scalings(Ab), golden["berlin_2018"]["scaling"])
    rng = np.random.default_rng(3)
    for _ in range(6):
        B = Ab.copy()
        B[rng.integers(0, B.shape[0]), rng.integers(0, B.shape[0])] += 1e-7
        with pytest.raises(RuntimeError, match="structure of calc_splines"):
            engine.les_scalings(B)
        with pytest.raises(RuntimeError, match="structure of calc_splines"):
            cs.scalings_from_les_matrix(B)
    with pytest.raises(RuntimeError, match="structure of calc_splines"):
        tph.iqp_handler.iqp_handler(ref, nv, np.eye(4 * n), 0.12, 3.4, False, False, 3.0)


def test_vel_profile_lateral_limit_starts_from_mean_friction():
    """tph.calc_vel_profile's fixed point for the lateral speed limit starts from the MEAN friction coefficient
    (ay_max_global = mean(mu) * min(ay_max)) and stops on a 0.5 % relative change -- with a non-uniform mu and a speed-dependent ggv
    the start shows in the result.  The shim and the oracle against the fixed point written out here (lateral limit only: straight-line
    limits far away), and against each other on a full profile."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_vel_profile as cv
    from oracle import vel_ref
    n = 400
    kappa = np.full(n, 0.02)                               # a circle of radius 50 m: no acceleration needed on a plateau
    el = np.full(n, 3.0)
    mu = np.where(np.arange(n) < n // 2, 0.4, 1.6)         # two long stretches of different friction
    v = np.arange(0.0, 90.1, 10.0)
    ggv = np.column_stack((v, np.full(v.size, 50.0), 14.0 - 0.12 * v))           # ay_max falls with speed
    axm = np.column_stack((v, np.full(v.size, 50.0)))
    radii = 1.0 / np.abs(kappa)

    def lateral(start_mu):
        vx = np.sqrt(start_mu * np.amin(ggv[:, 2]) * radii)
        for _ in range(100):
            vn = np.sqrt(mu * np.interp(vx, ggv[:, 0], ggv[:, 2]) * radii)
            done = np.max(np.abs(vn / vx - 1.0)) < 0.005
            vx = vn
            if done:
                break
        return vx
    want, other = lateral(float(np.mean(mu))), lateral(mu)
    inner = np.r_[60:140, 330:390]                          # well inside the two plateaus: the profile sits on the lateral limit
    assert np.min(np.abs(want - other)[inner]) > 1e-4       # the two starts are distinguishable there
    for f in (cv.calc_vel_profile, vel_ref.calc_vel_profile):
        got = f(ggv=ggv, ax_max_machines=axm, v_max=89.0, kappa=kappa, el_lengths=el, closed=True, mu=mu, drag_coeff=0.0, m_veh=1000.0)
        assert np.max(np.abs(got - want)[inner]) < 1e-9, f.__module__
    # full profile with binding acceleration limits: shim == oracle
    ggv2 = np.column_stack((v, 12.0 - 0.05 * v, 14.0 - 0.12 * v))
    axm2 = np.column_stack((v, np.interp(v, [0.0, 30.0, 90.0], [6.0, 5.0, 1.0])))
    a = cv.calc_vel_profile(ggv=ggv2, ax_max_machines=axm2, v_max=60.0, kappa=kappa, el_lengths=el, closed=True, mu=mu, drag_coeff=0.8, m_veh=1100.0)
    b = vel_ref.calc_vel_profile(ggv=ggv2, ax_max_machines=axm2, v_max=60.0, kappa=kappa, el_lengths=el, closed=True, mu=mu, drag_coeff=0.8, m_veh=1100.0)
    assert np.max(np.abs(a - b)) < 1e-9


def test_vel_profile_local_gg_and_unclosed_forms():
    """Round 6 (VERDICT r5 missing 4): the two forms of tph.calc_vel_profile's signature the shim had refused -- `loc_gg` [no_points, 2] (local
    ax_max / ay_max per point instead of the ggv diagram: what the mintime branch with a variable friction map passes [REF main_globaltraj.py:396-410])
    and unclosed profiles with v_start / v_end.  The shim (one gated pass with an `active` flag) against the oracle's restatement of upstream's
    work-list form, plus what can be said without either: a constant local gg equals the ggv form with constant rows; an unclosed profile starts at
    v_start, ends at or below v_end, never exceeds the lateral limit, and its accelerations respect the friction ellipse; the error texts."""
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_vel_profile as cv
    from oracle import vel_ref
    rng = np.random.default_rng(4)
    n = 300
    s = np.arange(n) * 3.0
    kappa = 0.03 * np.sin(2 * np.pi * s / 400.0) + 0.012 * np.sin(2 * np.pi * s / 90.0 + 1.0)
    el = np.full(n, 3.0)
    v = np.arange(0.0, 90.1, 10.0)
    axm = np.column_stack((v, np.interp(v, [0.0, 30.0, 90.0], [7.0, 5.0, 1.5])))
    loc = np.column_stack((10.0 + 3.0 * np.sin(s / 55.0), 12.0 + 4.0 * np.cos(s / 70.0)))
    kw = dict(ax_max_machines=axm, drag_coeff=0.85, m_veh=1160.0, dyn_model_exp=1.4)
    # closed, local gg
    a = cv.calc_vel_profile(kappa=kappa, el_lengths=el, closed=True, loc_gg=loc, v_max=70.0, **kw)
    b = vel_ref.calc_vel_profile(kappa=kappa, el_lengths=el, closed=True, loc_gg=loc, v_max=70.0, **kw)
    assert a.shape == (n,) and np.max(np.abs(a - b)) < 1e-9
    assert np.all(a <= np.minimum(np.sqrt(loc[:, 1] / np.abs(kappa)), 70.0) + 1e-9)
    # a constant local gg is the ggv form with constant rows
    const = np.column_stack((np.full(n, 9.0), np.full(n, 11.0)))
    ggv_c = np.column_stack((v, np.full(v.size, 9.0), np.full(v.size, 11.0)))
    c1 = cv.calc_vel_profile(kappa=kappa, el_lengths=el, closed=True, loc_gg=const, v_max=70.0, **kw)
    c2 = cv.calc_vel_profile(kappa=kappa, el_lengths=el, closed=True, ggv=ggv_c, v_max=70.0, **kw)
    assert np.max(np.abs(c1 - c2)) < 1e-9
    # unclosed: ggv form and local-gg form, with and without v_end, filter window included
    ggv = np.column_stack((v, 11.0 - 0.04 * v, 13.0 - 0.08 * v))
    mu = 0.8 + 0.4 * rng.uniform(size=n)
    for form in (dict(ggv=ggv, mu=mu, v_max=65.0), dict(loc_gg=loc, v_max=65.0)):
        for v_end in (None, 8.0):
            for fw in (None, 5):
                a = cv.calc_vel_profile(kappa=kappa, el_lengths=el[:-1], closed=False, v_start=12.0, v_end=v_end, filt_window=fw, **form, **kw)
                b = vel_ref.calc_vel_profile(kappa=kappa, el_lengths=el[:-1], closed=False, v_start=12.0, v_end=v_end, filt_window=fw, **form, **kw)
                assert a.shape == (n,) and np.max(np.abs(a - b)) < 1e-9, (sorted(form), v_end, fw)
                if fw is None:
                    assert a[0] <= 12.0 + 1e-12 and (v_end is None or a[-1] <= v_end + 1e-12)
                    ax = (a[1:] ** 2 - a[:-1] ** 2) / (2.0 * el[:-1])
                    assert np.max(ax) < 7.0 + 1e-9                  # never more than the machine gives
    # a negative start speed is taken as zero (upstream warns and goes on)
    z = cv.calc_vel_profile(kappa=kappa, el_lengths=el[:-1], closed=False, v_start=-3.0, ggv=ggv, v_max=65.0, **kw)
    assert z[0] == 0.0 and z[1] > 0.0
    for bad, msg in ((dict(closed=False, ggv=ggv, el_lengths=el[:-1]), "v_start must be provided"),
                     (dict(closed=True, loc_gg=loc, el_lengths=el), "v_max must be supplied if loc_gg is used"),
                     (dict(closed=True, loc_gg=loc[:-1], v_max=60.0, el_lengths=el), r"loc_gg must have the shape \[no_points, 2\]"),
                     (dict(closed=True, loc_gg=loc, ggv=ggv, v_max=60.0, el_lengths=el), "not both"),
                     (dict(closed=False, ggv=ggv, v_start=5.0, el_lengths=el), "el_lengths \\+ 1 if unclosed"),
                     (dict(closed=True, el_lengths=el), "Either ggv or loc_gg must be supplied")):
        for f in (cv.calc_vel_profile, vel_ref.calc_vel_profile):
            with pytest.raises(RuntimeError, match=msg):
                f(kappa=kappa, **bad, **kw)
