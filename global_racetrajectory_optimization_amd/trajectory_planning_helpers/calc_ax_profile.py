"""Host-side shim of tph.calc_ax_profile -- boundary [REF main_globaltraj.py:414-416]."""
import numpy as np


def calc_ax_profile(vx_profile: np.ndarray, el_lengths: np.ndarray, eq_length_output: bool = False) -> np.ndarray:
    if vx_profile.size != el_lengths.size + 1:
        raise RuntimeError("Array size of vx_profile should be 1 element bigger than el_lengths!")
    ax = (np.power(vx_profile[1:], 2) - np.power(vx_profile[:-1], 2)) / (2 * el_lengths)
    if eq_length_output:
        out = np.zeros(vx_profile.size)
        out[:-1] = ax
        return out
    return ax
