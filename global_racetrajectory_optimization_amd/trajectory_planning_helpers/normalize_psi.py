"""Host-side shim of tph.normalize_psi: wrap angles to [-pi, pi)."""
import math

import numpy as np


def normalize_psi(psi):
    psi_out = np.sign(psi) * np.mod(np.abs(psi), 2 * math.pi)
    if type(psi_out) is np.ndarray:
        psi_out[psi_out >= math.pi] -= 2 * math.pi
        psi_out[psi_out < -math.pi] += 2 * math.pi
    else:
        if psi_out >= math.pi:
            psi_out -= 2 * math.pi
        elif psi_out < -math.pi:
            psi_out += 2 * math.pi
    return psi_out
