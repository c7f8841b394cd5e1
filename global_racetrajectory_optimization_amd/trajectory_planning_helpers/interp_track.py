"""Host-side shim of tph.interp_track (linear re-sampling; used by spline_approximation)."""
import math

import numpy as np


def interp_track(track: np.ndarray, stepsize: float) -> np.ndarray:
    track_cl = np.vstack((track, track[0]))
    el = np.sqrt(np.sum(np.diff(track_cl[:, :2], axis=0) ** 2, axis=1))
    dists_cum = np.insert(np.cumsum(el), 0, 0.0)
    no_points = math.ceil(dists_cum[-1] / stepsize) + 1
    dists_interp = np.linspace(0.0, dists_cum[-1], no_points)
    out = np.column_stack([np.interp(dists_interp, dists_cum, track_cl[:, k]) for k in range(track_cl.shape[1])])
    return out[:-1]
