"""
Drop-in package named `trajectory_planning_helpers` for the reference's mincurv / mincurv_iqp flow.

The reference resolves `tph.opt_min_curv.opt_min_curv` [REF main_globaltraj.py:265] and `tph.iqp_handler.iqp_handler`
[REF main_globaltraj.py:274] at call time on `import trajectory_planning_helpers as tph` [REF main_globaltraj.py:6];
those two run on the MI355X engine (global_racetrajectory_optimization_amd.engine -> libmcq.so, hand-written HIP).
The other modules are thin host-side (numpy/scipy) shims of the pre/post-processing helpers that flow touches
 -- they exist because the real package is not installable in this image, and they are outside the
GPU hot path.
"""
import importlib as _importlib

_SUBMODULES = (
    "calc_splines", "calc_spline_lengths", "interp_splines", "create_raceline", "interp_track_widths", "interp_track",
    "side_of_line", "spline_approximation", "check_normals_crossing", "opt_min_curv", "iqp_handler",
    "opt_shortest_path", "import_veh_dyn_info", "calc_head_curv_an", "calc_vel_profile", "calc_ax_profile",
    "calc_t_profile", "progressbar", "calc_normal_vectors", "normalize_psi", "conv_filt",
)


def __getattr__(name):
    if name in _SUBMODULES:
        return _importlib.import_module("." + name, __name__)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
