"""
Runs the reference's main_globaltraj.py UNTOUCHED on top of the drop-in `trajectory_planning_helpers` package
(SURVEY.md App. C).  The script is an import-time program with user-editable literals [REF main_globaltraj.py:21-83];
the harness

  * copies the reference tree into a writable scratch directory (the script writes outputs/ next to itself
    [REF main_globaltraj.py:143]),
  * overrides ONLY literals inside the USER INPUT block (opt_type, plot switches, track name) textually and asserts
    that everything after that block is byte-identical to the reference,
  * pre-seeds sys.modules with our `trajectory_planning_helpers`, a stub `casadi` (imported, never called in the mincurv
    modes) and a no-op `pkg_resources.require` (the pinned numpy==1.18.1 etc. [REF requirements.txt:1-7] cannot be met),
  * runs it with matplotlib's Agg backend.

Usage:  python -m global_racetrajectory_optimization_amd.harness --reference /root/reference --opt-type mincurv
"""
import argparse
import hashlib
import importlib
import io
import os
import re
import shutil
import sys
import tempfile
import types

USER_BLOCK_END_MARK = "# CHECK USER INPUT"


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        return _Dummy()


def _stub_casadi():
    mod = types.ModuleType("casadi")
    mod.__getattr__ = lambda name: _Dummy()
    return mod


RECORDED_ENTRY_POINTS = (("opt_min_curv", "opt_min_curv"), ("iqp_handler", "iqp_handler"), ("opt_shortest_path", "opt_shortest_path"))


def _install_recorder(tph, record):
    """record (a list) receives one dict per call the script makes through the drop-in boundary [REF main_globaltraj.py:264-271, 273-284,
    286-290, 344-350]: the entry point's name and a deep copy of its keyword arguments as they were BEFORE the call (iqp_handler works on
    its arguments).  Returns the undo function."""
    import copy
    saved = []
    for mod_name, fn_name in RECORDED_ENTRY_POINTS:
        mod = getattr(tph, mod_name)
        orig = getattr(mod, fn_name)

        def wrapper(*args, _orig=orig, _name=mod_name + "." + fn_name, **kwargs):
            record.append(dict(entry=_name, args=copy.deepcopy(args), kwargs=copy.deepcopy(kwargs)))
            return _orig(*args, **kwargs)

        setattr(mod, fn_name, wrapper)
        saved.append((mod, fn_name, orig))

    def undo():
        for mod, fn_name, orig in saved:
            setattr(mod, fn_name, orig)
    return undo


def run(reference_dir, opt_type="mincurv", track_name="berlin_2018", scratch=None, quiet=False, record=None):
    reference_dir = os.path.abspath(reference_dir)
    scratch = scratch or tempfile.mkdtemp(prefix="globaltraj_")
    work = os.path.join(scratch, "reference")
    if os.path.exists(work):
        shutil.rmtree(work)
    shutil.copytree(reference_dir, work, ignore=shutil.ignore_patterns("__pycache__", "outputs", ".git"))

    src_path = os.path.join(work, "main_globaltraj.py")
    src = open(src_path).read()
    cut = src.index(USER_BLOCK_END_MARK)
    head, tail = src[:cut], src[cut:]
    tail_hash = hashlib.sha1(tail.encode()).hexdigest()

    head, n_sub = re.subn(r"(?m)^opt_type = ['\"][a-z_]+['\"]", "opt_type = '%s'" % opt_type, head)
    if n_sub != 1:
        raise RuntimeError("could not override opt_type inside the USER INPUT block")
    head, n_sub = re.subn(r'(?m)^file_paths\["track_name"\] = "[A-Za-z0-9_]+"',
                          'file_paths["track_name"] = "%s"' % track_name, head)
    if n_sub != 1:
        raise RuntimeError("could not override track_name inside the USER INPUT block")
    for key in ("mincurv_curv_lin", "raceline", "imported_bounds", "raceline_curv", "racetraj_vel", "racetraj_vel_3d",
                "spline_normals", "mintime_plots"):
        head = re.sub(r'"%s":\s*True' % key, '"%s": False' % key, head)
    new_src = head + tail
    assert hashlib.sha1(new_src[new_src.index(USER_BLOCK_END_MARK):].encode()).hexdigest() == tail_hash

    import matplotlib
    matplotlib.use("Agg")
    saved_path = list(sys.path)
    saved_modules = {k: sys.modules.get(k) for k in ("trajectory_planning_helpers", "casadi")}
    sys.path.insert(0, work)
    captured = None
    try:
        tph = importlib.import_module("global_racetrajectory_optimization_amd.trajectory_planning_helpers")
        sys.modules["trajectory_planning_helpers"] = tph
        for sub in tph._SUBMODULES:
            sys.modules["trajectory_planning_helpers." + sub] = getattr(tph, sub)
        undo_recorder = _install_recorder(tph, record) if record is not None else (lambda: None)
        if "casadi" not in sys.modules:
            sys.modules["casadi"] = _stub_casadi()
        import pkg_resources
        orig_require = pkg_resources.require
        pkg_resources.require = lambda *a, **k: []
        glb = {"__name__": "__main__", "__file__": src_path}
        cwd = os.getcwd()
        os.chdir(work)
        stdout = sys.stdout
        if quiet:
            sys.stdout = io.StringIO()
        try:
            exec(compile(new_src, src_path, "exec"), glb)
        finally:
            if quiet:
                captured = sys.stdout.getvalue()
            sys.stdout = stdout
            os.chdir(cwd)
            pkg_resources.require = orig_require
            undo_recorder()
    finally:
        sys.path[:] = saved_path
        for k, v in saved_modules.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k in [m for m in sys.modules if m.startswith(("helper_funcs_glob", "opt_mintime_traj", "frictionmap",
                                                          "trajectory_planning_helpers."))]:
            sys.modules.pop(k, None)
    return dict(outputs=os.path.join(work, "outputs", "traj_race_cl.csv"), globals=glb, stdout=captured, workdir=work)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--opt-type", default="mincurv", choices=("mincurv", "mincurv_iqp", "shortest_path"))
    ap.add_argument("--track", default="berlin_2018")
    args = ap.parse_args()
    res = run(args.reference, args.opt_type, args.track)
    print("HARNESS: wrote", res["outputs"], "exists:", os.path.exists(res["outputs"]))


if __name__ == "__main__":
    main()
