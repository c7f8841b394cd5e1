"""Host-side shim of tph.interp_track_widths (inside the IQP loop)."""
import numpy as np


def interp_track_widths(w_track: np.ndarray, spline_inds: np.ndarray, t_values: np.ndarray,
                        incl_last_point: bool = False) -> np.ndarray:
    w_cl = np.vstack((w_track, w_track[0]))
    m = t_values.size
    out = np.zeros((m, w_track.shape[1]))
    k = m - 1 if incl_last_point else m
    lo = w_cl[spline_inds[:k]]
    hi = w_cl[spline_inds[:k] + 1]
    out[:k] = lo + (hi - lo) * t_values[:k, None]
    if incl_last_point:
        out[-1] = w_cl[-1]
    return out
