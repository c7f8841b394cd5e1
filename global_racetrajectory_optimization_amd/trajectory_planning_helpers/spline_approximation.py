"""
Host-side shim of tph.spline_approximation.spline_approximation -- boundary
[REF helper_funcs_glob/src/prep_track.py:39-45].  FITPACK smoothing (scipy) + per-point closest-parameter search;
host only, not part of the GPU hot path.
"""
import math

import numpy as np
from scipy import interpolate, optimize

from . import interp_track as _it
from . import side_of_line as _sol


def _dist_to_p(t_glob, tck, p):
    s = np.asarray(interpolate.splev(t_glob, tck)).reshape(2)
    return math.hypot(s[0] - p[0], s[1] - p[1])


def spline_approximation(track: np.ndarray, k_reg: int = 3, s_reg: int = 10, stepsize_prep: float = 1.0,
                         stepsize_reg: float = 3.0, debug: bool = False) -> np.ndarray:
    track_interp = _it.interp_track(track=track, stepsize=stepsize_prep)
    track_interp_cl = np.vstack((track_interp, track_interp[0]))

    track_cl = np.vstack((track, track[0]))
    n_cl = track_cl.shape[0]
    el_cl = np.sqrt(np.sum(np.diff(track_cl[:, :2], axis=0) ** 2, axis=1))
    dists_cum_cl = np.insert(np.cumsum(el_cl), 0, 0.0)

    tck_cl = interpolate.splprep([track_interp_cl[:, 0], track_interp_cl[:, 1]], k=k_reg, s=s_reg, per=1)[0]

    no_points_lencalc = math.ceil(dists_cum_cl[-1]) * 4
    path_tmp = np.array(interpolate.splev(np.linspace(0.0, 1.0, no_points_lencalc), tck_cl)).T
    len_smoothed = float(np.sum(np.sqrt(np.sum(np.diff(path_tmp, axis=0) ** 2, axis=1))))

    no_points_reg_cl = math.ceil(len_smoothed / stepsize_reg) + 1
    path_smoothed = np.array(interpolate.splev(np.linspace(0.0, 1.0, no_points_reg_cl), tck_cl)).T[:-1]

    dists_closest = np.zeros(n_cl)
    closest_point = np.zeros((n_cl, 2))
    closest_t = np.zeros(n_cl)
    t_guess = dists_cum_cl / dists_cum_cl[-1]
    for i in range(n_cl):
        closest_t[i] = optimize.fmin(_dist_to_p, x0=t_guess[i], args=(tck_cl, track_cl[i, :2]), disp=False)[0]
        closest_point[i] = np.asarray(interpolate.splev(closest_t[i], tck_cl)).reshape(2)
        dists_closest[i] = math.hypot(closest_point[i, 0] - track_cl[i, 0], closest_point[i, 1] - track_cl[i, 1])
    if debug:
        print("Spline approximation: mean deviation %.2fm, maximum deviation %.2fm"
              % (float(np.mean(dists_closest)), float(np.amax(np.abs(dists_closest)))))

    sides = np.zeros(n_cl - 1)
    for i in range(n_cl - 1):
        sides[i] = _sol.side_of_line(a=track_cl[i, :2], b=track_cl[i + 1, :2], z=closest_point[i])
    sides_cl = np.hstack((sides, sides[0]))

    w_right_new = track_cl[:, 2] + sides_cl * dists_closest
    w_left_new = track_cl[:, 3] - sides_cl * dists_closest

    grid = np.linspace(0.0, 1.0, no_points_reg_cl)
    w_right_s = np.interp(grid, closest_t, w_right_new)
    w_left_s = np.interp(grid, closest_t, w_left_new)
    return np.column_stack((path_smoothed, w_right_s[:-1], w_left_s[:-1]))
