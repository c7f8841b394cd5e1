"""BASELINE config 1: the reference's main_globaltraj.py runs UNTOUCHED on top of the drop-in package (harness notes:
SURVEY.md App. C).  Needs the reference checkout (/root/reference) -> skipped on the GPU box.  The engine behind
opt_min_curv here is the SIMT-interpreted kernel library (test infrastructure); the `-m gpu` suite covers the real one."""
import os

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_dir():
    """The reference checkout: /root/reference in the build container; on the GPU box only when a round script shipped a scratch
    copy along (scratch_ft/ is git-ignored: it travels with a gpurun snapshot and is deleted afterwards, never committed)."""
    for d in (os.environ.get("MCQ_REFERENCE_DIR"), REF, os.path.join(ROOT, "scratch_ft", "reference")):
        if d and os.path.exists(os.path.join(d, "main_globaltraj.py")):
            return d
    return None


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main_globaltraj.py")), reason="reference checkout not present")
@pytest.mark.parametrize("opt_type", ["mincurv", "mincurv_iqp", "shortest_path"])
def test_main_globaltraj_untouched(emu_lib, tmp_path, monkeypatch, opt_type):
    from global_racetrajectory_optimization_amd import engine, harness
    monkeypatch.setenv("MCQ_LIB", emu_lib)
    monkeypatch.setattr(engine, "_DEFAULT_ENGINE", None)
    res = harness.run(REF, opt_type=opt_type, track_name="rounded_rectangle", scratch=str(tmp_path), quiet=True)
    monkeypatch.setattr(engine, "_DEFAULT_ENGINE", None)
    assert "INFO: Estimated laptime:" in res["stdout"]
    marker = {"mincurv": "Solver runtime opt_min_curv", "mincurv_iqp": "Minimum curvature IQP: iteration 3",
              "shortest_path": "Solver runtime opt_shortest_path"}[opt_type]
    assert marker in res["stdout"]
    csv = res["outputs"]
    assert os.path.exists(csv)
    lines = open(csv).read().splitlines()
    assert lines[2].replace(" ", "") == "#s_m;x_m;y_m;psi_rad;kappa_radpm;vx_mps;ax_mps2"
    data = np.loadtxt(csv, comments="#", delimiter=";")
    assert data.shape[1] == 7 and data.shape[0] > 100
    assert np.allclose(data[0, 1:3], data[-1, 1:3])          # closed: last row repeats the first point
    # the alpha the script got from the drop-in equals the golden single-shot solution
    if opt_type == "mincurv":
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rounded_rectangle.npz"))
        assert np.max(np.abs(res["globals"]["alpha_opt"] - g["alpha"])) < 1e-8
    elif opt_type == "mincurv_iqp":
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rounded_rectangle.npz"))
        assert np.max(np.abs(res["globals"]["alpha_opt"] - g["iqp_alpha"])) < 1e-7
    else:
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", "shortest_path.npz"))
        assert np.max(np.abs(res["globals"]["alpha_opt"] - z["rounded_rectangle_alpha"])) < 1e-8


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main_globaltraj.py")), reason="reference checkout not present")
def test_main_globaltraj_untouched_on_the_goldfarb_idnani_path(emu_lib, tmp_path, monkeypatch):
    """$MCQ_ALGORITHM=gi: the untouched script with EVERY QP of its mincurv_iqp flow solved by the engine's Goldfarb-Idnani path -- the algorithm of
    the quadprog it replaces -- ends in the golden IQP state of the default path."""
    from global_racetrajectory_optimization_amd import engine, harness
    monkeypatch.setenv("MCQ_LIB", emu_lib)
    monkeypatch.setenv("MCQ_ALGORITHM", "gi")
    monkeypatch.setattr(engine, "_DEFAULT_ENGINE", None)
    res = harness.run(REF, opt_type="mincurv_iqp", track_name="rounded_rectangle", scratch=str(tmp_path), quiet=True)
    monkeypatch.setattr(engine, "_DEFAULT_ENGINE", None)
    assert "INFO: Estimated laptime:" in res["stdout"] and "Minimum curvature IQP: iteration 3" in res["stdout"]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rounded_rectangle.npz"))
    assert np.max(np.abs(res["globals"]["alpha_opt"] - g["iqp_alpha"])) < 1e-7


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main_globaltraj.py")), reason="reference checkout not present")
def test_main_globaltraj_untouched_berlin_config1(emu_lib, tmp_path, monkeypatch):
    """BASELINE config 1 as specified: berlin_2018 (N = 776 at the ini defaults), opt_type = 'mincurv', the untouched script end to
    end -- import_track, prep_track, opt_min_curv behind the boundary, create_raceline, velocity profile, lap time, export."""
    from global_racetrajectory_optimization_amd import engine, harness
    monkeypatch.setenv("MCQ_LIB", emu_lib)
    monkeypatch.setattr(engine, "_DEFAULT_ENGINE", None)
    res = harness.run(REF, opt_type="mincurv", track_name="berlin_2018", scratch=str(tmp_path), quiet=True)
    monkeypatch.setattr(engine, "_DEFAULT_ENGINE", None)
    assert "Solver runtime opt_min_curv" in res["stdout"] and "INFO: Estimated laptime:" in res["stdout"]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "berlin_2018.npz"))
    assert res["globals"]["reftrack_interp"].shape == (776, 4)
    assert np.max(np.abs(res["globals"]["reftrack_interp"] - g["reftrack"])) < 1e-9      # same prep as the golden inputs
    assert np.max(np.abs(res["globals"]["alpha_opt"] - g["alpha"])) < 1e-6
    data = np.loadtxt(res["outputs"], comments="#", delimiter=";")
    assert data.shape[1] == 7 and np.allclose(data[0, 1:3], data[-1, 1:3])
    lap = float([l for l in res["stdout"].splitlines() if "Estimated laptime" in l][0].split(":")[-1].strip().rstrip("s"))
    # the lap time of the golden raceline through the ORACLE's velocity-profile chain (oracle/vel_ref.py)
    from oracle import tph_ref, vel_ref
    from global_racetrajectory_optimization_amd.trajectory_planning_helpers import calc_head_curv_an as ch
    out = tph_ref.create_raceline(g["reftrack"][:, :2], g["normvec"], g["alpha"], 2.0)
    _, kappa = ch.calc_head_curv_an(coeffs_x=out[2], coeffs_y=out[3], ind_spls=out[4], t_spls=out[5])
    gl = res["globals"]
    vx = vel_ref.calc_vel_profile(ax_max_machines=gl["ax_max_machines"], kappa=kappa, el_lengths=out[8], closed=True,
                                  drag_coeff=gl["pars"]["veh_params"]["dragcoeff"], m_veh=gl["pars"]["veh_params"]["mass"],
                                  ggv=gl["ggv"], v_max=gl["pars"]["veh_params"]["v_max"],
                                  dyn_model_exp=gl["pars"]["vel_calc_opts"]["dyn_model_exp"])
    assert abs(lap - vel_ref.lap_time_stable(vx, out[8])) < 0.02          # the script prints two decimals


@pytest.mark.gpu
@pytest.mark.parametrize("opt_type", ["mincurv", "mincurv_iqp"])
def test_main_globaltraj_untouched_on_the_gpu(tmp_path, monkeypatch, opt_type):
    """Config 1 on the REAL library: the untouched main_globaltraj.py drives libmcq.so on the MI355X (Berlin, ini defaults).  Needs the
    reference tree next to the repo snapshot (scripts/gpu_with_reference.sh ships a scratch copy); skipped otherwise."""
    ref = _reference_dir()
    if ref is None:
        pytest.skip("reference checkout not reachable on this box")
    from global_racetrajectory_optimization_amd import engine, harness
    monkeypatch.delenv("MCQ_LIB", raising=False)
    monkeypatch.setattr(engine, "_DEFAULT_ENGINE", None)
    res = harness.run(ref, opt_type=opt_type, track_name="berlin_2018", scratch=str(tmp_path), quiet=True)
    monkeypatch.setattr(engine, "_DEFAULT_ENGINE", None)
    out = res["stdout"]
    print(out)
    assert "INFO: Estimated laptime:" in out
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "berlin_2018.npz"))
    if opt_type == "mincurv":
        assert "Solver runtime opt_min_curv" in out
        assert np.max(np.abs(res["globals"]["alpha_opt"] - g["alpha"])) < 1e-6
    else:
        assert "Minimum curvature IQP: iteration 3" in out and "Finished IQP!" in out
        assert res["globals"]["alpha_opt"].shape[0] == res["globals"]["reftrack_interp"].shape[0]     # main rebinds both [REF :274]
    data = np.loadtxt(res["outputs"], comments="#", delimiter=";")
    assert data.shape[1] == 7 and data.shape[0] > 1000 and np.allclose(data[0, 1:3], data[-1, 1:3])
