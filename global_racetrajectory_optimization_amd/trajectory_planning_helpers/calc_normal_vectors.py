"""Host-side shim of tph.calc_normal_vectors (plots only, [REF helper_funcs_glob/src/result_plots.py:37-38])."""
import math

import numpy as np


def calc_normal_vectors(psi: np.ndarray) -> np.ndarray:
    psi_ = np.asarray(psi, dtype=np.float64) + math.pi / 2      # tangent direction in the maths convention
    tang = np.column_stack((np.cos(psi_), np.sin(psi_)))
    return np.column_stack((tang[:, 1], -tang[:, 0]))           # rotated clockwise: pointing right
