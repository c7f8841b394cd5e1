"""Host-side shim of tph.calc_spline_lengths (used by create_raceline)."""
import numpy as np


def calc_spline_lengths(coeffs_x: np.ndarray, coeffs_y: np.ndarray, quickndirty: bool = False,
                        no_interp_points: int = 15) -> np.ndarray:
    coeffs_x = np.atleast_2d(coeffs_x)
    coeffs_y = np.atleast_2d(coeffs_y)
    if coeffs_x.shape[0] != coeffs_y.shape[0]:
        raise RuntimeError("Coefficient matrices must have the same length!")
    if quickndirty:
        return np.sqrt(np.sum(coeffs_x[:, 1:], axis=1) ** 2 + np.sum(coeffs_y[:, 1:], axis=1) ** 2)
    t = np.linspace(0.0, 1.0, no_interp_points)
    tpow = np.stack((np.ones_like(t), t, t * t, t * t * t))
    px = coeffs_x @ tpow
    py = coeffs_y @ tpow
    return np.sum(np.hypot(np.diff(px, axis=1), np.diff(py, axis=1)), axis=1)
