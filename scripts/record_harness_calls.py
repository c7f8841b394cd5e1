"""
Records what the reference's UNTOUCHED main_globaltraj.py hands the drop-in boundary on BASELINE config 1 / 2 (berlin_2018, ini defaults), so
that the GPU box -- which has no reference tree -- can replay exactly those calls through the drop-in package on libmcq.so
(tests/test_harness.py::test_replay_of_the_recorded_calls_on_the_gpu).

Run in the build container (needs /root/reference):   python scripts/record_harness_calls.py
Writes tests/golden/harness_calls_berlin.npz:

  <call>_reftrack / _normvectors / _kwargs(json)      the keyword arguments of the call as the script made it
        mincurv   [REF main_globaltraj.py:264-271]     tph.opt_min_curv.opt_min_curv(reftrack=, normvectors=, A=, kappa_bound=, w_veh=, print_debug=, plot_debug=)[0]
        iqp       [REF main_globaltraj.py:273-284]     tph.iqp_handler.iqp_handler(..., stepsize_interp=, iters_min=, curv_error_allowed=)
        shortest  [REF main_globaltraj.py:286-290]     tph.opt_shortest_path.opt_shortest_path(reftrack=, normvectors=, w_veh=, print_debug=)
        reopt     [REF main_globaltraj.py:337-350]     the re-optimisation call of the mintime branch.  That branch needs casadi/IPOPT (absent, out of
                                                       scope), so this ONE record is constructed, not captured: the same statements applied to the
                                                       reference line instead of a mintime raceline (corridor 0.5 w_tr_reopt either side, w_veh_reopt).
  A is not stored (3104 x 3104 doubles): the script builds it with tph.calc_splines.calc_splines(path=closed reference line)
  [REF helper_funcs_glob/src/prep_track.py:48-51]; the recorder checks that rebuilding it that way from the recorded reftrack gives the SAME
  matrix bit for bit and stores its SHA-256 and the N spline scalings it carries (the one thing the engine reads from it); the replay rebuilds
  it and checks both.
  <call>_oracle_*          the ORACLE's outputs for those inputs (oracle/tph_ref.py + oracle/gi_dense.c: dense 4N x 4N inverse, all 4N rows)
  <call>_stdout            the lines the drop-in printed under print_debug (format only: the runtime differs)
  error cases              inputs the reference's callers can produce that must map to upstream's exceptions: a corridor narrower than the
                           vehicle -> RuntimeError("Problem not solvable, ..."), an unreachable curvature bound -> quadprog's
                           ValueError("constraints are inconsistent, no solution") -- with the oracle's own verdict recorded next to them.

The engine under the script while recording is the SIMT-interpreted kernel library (tests/emu); its outputs are NOT stored.
PARITY UNPINNED by the reference (no tph / quadprog here): the expected outputs are our oracle's.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF = "/root/reference"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def main():
    from global_racetrajectory_optimization_amd import engine, harness
    from global_racetrajectory_optimization_amd import trajectory_planning_helpers as tph
    from oracle import tph_ref
    import subprocess
    emu = os.path.join(ROOT, "tests", "emu", "libmcq_emu.so")
    if not os.path.exists(emu):
        subprocess.run([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], check=True)
    os.environ["MCQ_LIB"] = emu
    out = {}
    calls = {}
    for opt_type, key in (("mincurv", "mincurv"), ("mincurv_iqp", "iqp"), ("shortest_path", "shortest")):
        rec = []
        engine._DEFAULT_ENGINE = None
        res = harness.run(REF, opt_type=opt_type, track_name="berlin_2018", scratch=tempfile.mkdtemp(prefix="rec_"), quiet=True, record=rec)
        engine._DEFAULT_ENGINE = None
        assert len(rec) == 1 and not rec[0]["args"], (opt_type, len(rec))
        kw = rec[0]["kwargs"]
        calls[key] = kw
        out[key + "_entry"] = rec[0]["entry"]
        out[key + "_reftrack"] = kw["reftrack"]
        out[key + "_normvectors"] = kw["normvectors"]
        scal = {k: v for k, v in kw.items() if k not in ("reftrack", "normvectors", "A")}
        out[key + "_kwargs"] = json.dumps(scal, sort_keys=True)
        if "A" in kw:
            A2 = tph.calc_splines.calc_splines(path=np.vstack((kw["reftrack"][:, :2], kw["reftrack"][0, :2])))[2]
            assert np.array_equal(A2, kw["A"]), "A is not calc_splines(path=closed reference line)"
            out[key + "_A_sha256"] = sha(kw["A"])
            out[key + "_A_scalings"] = tph.calc_splines.scalings_from_les_matrix(kw["A"])
        want = {"mincurv": ("Solver runtime opt_min_curv",), "iqp": ("Minimum curvature IQP: iteration", "Finished IQP!"),
                "shortest": ("Solver runtime opt_shortest_path",)}[key]
        out[key + "_stdout"] = json.dumps([l for l in res["stdout"].splitlines() if l.startswith(want)])
        print(key, rec[0]["entry"], scal, out[key + "_stdout"], flush=True)

    # ---- the oracle's outputs ----------------------------------------------------------------------------------------------
    kw = calls["mincurv"]
    a, err = tph_ref.opt_min_curv(kw["reftrack"], kw["normvectors"], kw["A"], kw["kappa_bound"], kw["w_veh"])
    out["mincurv_oracle_alpha"], out["mincurv_oracle_curv_error_max"] = a, err
    g = np.load(os.path.join(ROOT, "tests", "golden", "berlin_2018.npz"))
    print("mincurv: oracle vs tests/golden/berlin_2018.npz %.2e m (same inputs: %s)" % (
        np.max(np.abs(a - g["alpha"])), np.array_equal(kw["reftrack"], g["reftrack"])), flush=True)

    kw = calls["iqp"]
    trace = []
    a, rt, nv = tph_ref.iqp_handler(kw["reftrack"], kw["normvectors"], kw["A"], kw["kappa_bound"], kw["w_veh"], kw["stepsize_interp"],
                                    iters_min=kw["iters_min"], curv_error_allowed=kw["curv_error_allowed"], trace=trace)
    out["iqp_oracle_alpha"], out["iqp_oracle_reftrack"], out["iqp_oracle_normvectors"] = a, rt, nv
    out["iqp_oracle_curv_error_trace"] = np.array([t["curv_error_max"] for t in trace])
    out["iqp_oracle_n_trace"] = np.array([t["n"] for t in trace])
    print("iqp: %d passes, n %s, curv errors %s" % (len(trace), [t["n"] for t in trace], ["%.4f" % t["curv_error_max"] for t in trace]), flush=True)

    kw = calls["shortest"]
    out["shortest_oracle_alpha"] = tph_ref.opt_shortest_path(kw["reftrack"], kw["normvectors"], kw["w_veh"])

    # ---- the re-optimisation call of the mintime branch, constructed [REF main_globaltraj.py:337-350; params/racecar.ini:110-111] ----------
    base = calls["mincurv"]
    w_tr_reopt, w_veh_reopt = 2.0, 1.6
    w_tr_tmp = 0.5 * w_tr_reopt * np.ones(base["reftrack"].shape[0])
    reopt_ref = np.column_stack((base["reftrack"][:, :2], w_tr_tmp, w_tr_tmp))
    out["reopt_reftrack"], out["reopt_normvectors"] = reopt_ref, base["normvectors"]
    out["reopt_kwargs"] = json.dumps(dict(kappa_bound=base["kappa_bound"], w_veh=w_veh_reopt, print_debug=base["print_debug"], plot_debug=False),
                                     sort_keys=True)
    out["reopt_A_sha256"] = sha(base["A"])
    out["reopt_A_scalings"] = tph.calc_splines.scalings_from_les_matrix(base["A"])
    a, err = tph_ref.opt_min_curv(reopt_ref, base["normvectors"], base["A"], base["kappa_bound"], w_veh_reopt)
    out["reopt_oracle_alpha"], out["reopt_oracle_curv_error_max"] = a, err

    # ---- error mapping: what upstream's callers would see ------------------------------------------------------------------------------
    narrow = base["reftrack"].copy()
    narrow[100:110, 2:] = 1.5                      # w_r + w_l = 3.0 < w_veh = 3.4
    try:
        tph_ref.opt_min_curv(narrow, base["normvectors"], base["A"], base["kappa_bound"], base["w_veh"])
        raise AssertionError("the oracle accepted a corridor narrower than the vehicle")
    except RuntimeError as e:
        out["error_narrow_message"] = str(e)
    out["error_narrow_rows"] = np.arange(100, 110)
    out["error_narrow_width"] = 1.5
    kb_bad = 0.004                                 # below what any line inside the corridor can reach on Berlin's hairpins
    try:
        tph_ref.opt_min_curv(base["reftrack"], base["normvectors"], base["A"], kb_bad, base["w_veh"])
        raise AssertionError("the oracle found the curvature bound reachable")
    except ValueError as e:
        out["error_kappa_message"] = str(e)
    out["error_kappa_bound"] = kb_bad
    print("errors:", out["error_narrow_message"], "|", out["error_kappa_message"], flush=True)

    path = os.path.join(ROOT, "tests", "golden", "harness_calls_berlin.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
