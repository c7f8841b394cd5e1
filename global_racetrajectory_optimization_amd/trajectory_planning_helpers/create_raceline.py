"""Host-side shim of tph.create_raceline.create_raceline -- boundary [REF main_globaltraj.py:371-376] (9-tuple)."""
import numpy as np

from . import calc_spline_lengths as _csl
from . import calc_splines as _cs
from . import interp_splines as _is


def create_raceline(refline: np.ndarray, normvectors: np.ndarray, alpha: np.ndarray, stepsize_interp: float) -> tuple:
    raceline = refline + np.expand_dims(alpha, 1) * normvectors
    raceline_cl = np.vstack((raceline, raceline[0]))
    coeffs_x, coeffs_y, A_raceline, _ = _cs.calc_splines(path=raceline_cl, use_dist_scaling=False)
    spline_lengths = _csl.calc_spline_lengths(coeffs_x=coeffs_x, coeffs_y=coeffs_y)
    raceline_interp, spline_inds, t_values, s_interp = _is.interp_splines(
        spline_lengths=spline_lengths, coeffs_x=coeffs_x, coeffs_y=coeffs_y, incl_last_point=False,
        stepsize_approx=stepsize_interp)
    s_tot = float(np.sum(spline_lengths))
    el_lengths_cl = np.append(np.diff(s_interp), s_tot - s_interp[-1])
    return (raceline_interp, A_raceline, coeffs_x, coeffs_y, spline_inds, t_values, s_interp, spline_lengths,
            el_lengths_cl)
