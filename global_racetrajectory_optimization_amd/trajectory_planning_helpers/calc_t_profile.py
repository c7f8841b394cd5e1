"""Host-side shim of tph.calc_t_profile -- boundary [REF main_globaltraj.py:419-421]; lap time = last entry."""
import math

import numpy as np

from . import calc_ax_profile as _ax


def calc_t_profile(vx_profile: np.ndarray, el_lengths: np.ndarray, t_start: float = 0.0,
                   ax_profile: np.ndarray = None) -> np.ndarray:
    if vx_profile.size < el_lengths.size:
        raise RuntimeError("vx_profile and el_lenghts must have at least the same length!")
    if ax_profile is not None and ax_profile.size < el_lengths.size:
        raise RuntimeError("ax_profile and el_lenghts must have at least the same length!")
    if ax_profile is None:
        ax_profile = _ax.calc_ax_profile(vx_profile=vx_profile, el_lengths=el_lengths, eq_length_output=False)
    no = el_lengths.size
    t_steps = np.zeros(no)
    for i in range(no):
        if not math.isclose(ax_profile[i], 0.0):
            t_steps[i] = (-vx_profile[i] + math.sqrt(vx_profile[i] ** 2 + 2 * ax_profile[i] * el_lengths[i])) / ax_profile[i]
        else:
            t_steps[i] = el_lengths[i] / vx_profile[i]
    return np.insert(np.cumsum(t_steps), 0, 0.0) + t_start
