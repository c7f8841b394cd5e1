"""Host-side shim of tph.interp_splines (used by create_raceline)."""
import math

import numpy as np


def interp_splines(coeffs_x: np.ndarray, coeffs_y: np.ndarray, spline_lengths: np.ndarray = None,
                   incl_last_point: bool = False, stepsize_approx: float = None, stepnum_fixed: list = None) -> tuple:
    if stepnum_fixed is not None:
        raise NotImplementedError("interp_splines shim: stepnum_fixed is not used on the mincurv flow")
    if stepsize_approx is None:
        raise RuntimeError("Provide one of 'stepsize_approx' and 'stepnum_fixed' and set the other to 'None'!")
    coeffs_x = np.atleast_2d(coeffs_x)
    coeffs_y = np.atleast_2d(coeffs_y)
    if spline_lengths is None:
        from . import calc_spline_lengths as csl
        spline_lengths = csl.calc_spline_lengths(coeffs_x, coeffs_y, quickndirty=False)
    dists_cum = np.cumsum(spline_lengths)
    no_interp_points = math.ceil(dists_cum[-1] / stepsize_approx) + 1
    dists_interp = np.linspace(0.0, dists_cum[-1], no_interp_points)
    m = no_interp_points - 1
    q = dists_interp[:m]
    inds = np.searchsorted(dists_cum, q, side="right")          # first j with q < dists_cum[j]
    inds = np.minimum(inds, dists_cum.size - 1)
    start = np.where(inds > 0, dists_cum[np.maximum(inds - 1, 0)], 0.0)
    t = (q - start) / spline_lengths[inds]
    path = np.empty((m, 2))
    path[:, 0] = coeffs_x[inds, 0] + t * (coeffs_x[inds, 1] + t * (coeffs_x[inds, 2] + t * coeffs_x[inds, 3]))
    path[:, 1] = coeffs_y[inds, 0] + t * (coeffs_y[inds, 1] + t * (coeffs_y[inds, 2] + t * coeffs_y[inds, 3]))
    if incl_last_point:
        path = np.vstack((path, [np.sum(coeffs_x[-1]), np.sum(coeffs_y[-1])]))
        inds = np.append(inds, coeffs_x.shape[0] - 1)
        t = np.append(t, 1.0)
        return path, inds, t, dists_interp
    return path, inds, t, dists_interp[:-1]
