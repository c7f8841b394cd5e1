"""Host-side shim of tph.calc_head_curv_an -- boundary [REF main_globaltraj.py:383-387]; heading 0 = north."""
import math

import numpy as np

from . import normalize_psi as _np


def calc_head_curv_an(coeffs_x: np.ndarray, coeffs_y: np.ndarray, ind_spls: np.ndarray, t_spls: np.ndarray,
                      calc_curv: bool = True, calc_dcurv: bool = False) -> tuple:
    if coeffs_x.shape[0] != coeffs_y.shape[0]:
        raise ValueError("Coefficient matrices must have the same length!")
    if ind_spls.size != t_spls.size:
        raise ValueError("ind_spls and t_spls must have the same length!")
    cx, cy, t = coeffs_x[ind_spls], coeffs_y[ind_spls], t_spls
    x_d = cx[:, 1] + 2 * cx[:, 2] * t + 3 * cx[:, 3] * t ** 2
    y_d = cy[:, 1] + 2 * cy[:, 2] * t + 3 * cy[:, 3] * t ** 2
    psi = _np.normalize_psi(np.arctan2(y_d, x_d) - math.pi / 2)
    if not calc_curv:
        return psi
    x_dd = 2 * cx[:, 2] + 6 * cx[:, 3] * t
    y_dd = 2 * cy[:, 2] + 6 * cy[:, 3] * t
    kappa = (x_d * y_dd - y_d * x_dd) / np.power(x_d ** 2 + y_d ** 2, 1.5)
    if not calc_dcurv:
        return psi, kappa
    x_ddd, y_ddd = 6 * cx[:, 3], 6 * cy[:, 3]
    dkappa = ((x_d ** 2 + y_d ** 2) * (x_d * y_ddd - y_d * x_ddd) - 3 * (x_d * y_dd - y_d * x_dd) * (x_d * x_dd + y_d * y_dd)) \
        / np.power(x_d ** 2 + y_d ** 2, 3)
    return psi, kappa, dkappa
