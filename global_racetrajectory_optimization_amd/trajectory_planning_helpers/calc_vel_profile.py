"""
Host-side shim of tph.calc_vel_profile -- boundary [REF main_globaltraj.py:400-410, 469-479].

Forward/backward quasi-steady-state velocity profile over the ggv diagram (a "next" row, host
side for now, outside the GPU hot path).  Lateral limit v = sqrt(ay_max(v) * R) by fixed-point iteration, then an
acceleration-limited forward sweep and a deceleration-limited backward sweep; closed tracks are swept over two laps so
that the result is periodic.
"""
import numpy as np

from . import conv_filt as _cf


def _ax_possible(v, radius, ggv, ax_max_machines, mode, dyn_model_exp, drag_coeff, m_veh, mu=1.0):
    ax_tires = mu * np.interp(v, ggv[:, 0], ggv[:, 1])
    ay_tires = mu * np.interp(v, ggv[:, 0], ggv[:, 2])
    ay_used = v * v / radius
    radicand = 1.0 - (ay_used / ay_tires) ** dyn_model_exp if ay_tires > 0.0 else 0.0
    ax_avail_tires = ax_tires * radicand ** (1.0 / dyn_model_exp) if radicand > 0.0 else 0.0
    ax_drag = -v * v * drag_coeff / m_veh
    if mode == "accel":
        ax_machine = np.interp(v, ax_max_machines[:, 0], ax_max_machines[:, 1])
        return min(ax_avail_tires, ax_machine) + ax_drag
    return ax_avail_tires - ax_drag        # braking, integrated backwards


def _sweep(vx, radii, el, ggv, ax_max_machines, mode, dyn_model_exp, drag_coeff, m_veh):
    v = vx.copy()
    for i in range(v.size - 1):
        ax = _ax_possible(v[i], radii[i], ggv, ax_max_machines, mode, dyn_model_exp, drag_coeff, m_veh)
        v_next = np.sqrt(max(v[i] * v[i] + 2.0 * ax * el[i], 0.0))
        if v_next < v[i + 1]:
            v[i + 1] = v_next
    return v


def calc_vel_profile(ax_max_machines: np.ndarray, kappa: np.ndarray, el_lengths: np.ndarray, closed: bool,
                     drag_coeff: float, m_veh: float, ggv: np.ndarray = None, loc_gg: np.ndarray = None,
                     v_max: float = None, dyn_model_exp: float = 1.0, mu: np.ndarray = None, v_start: float = None,
                     v_end: float = None, filt_window: int = None) -> np.ndarray:
    if ggv is None or loc_gg is not None or mu is not None:
        raise NotImplementedError("calc_vel_profile shim: only the global-ggv form used by main_globaltraj.py")
    if not closed:
        raise NotImplementedError("calc_vel_profile shim: closed tracks only")
    if kappa.size != el_lengths.size:
        raise RuntimeError("kappa and el_lengths must have the same length if closed!")
    if not 1.0 <= dyn_model_exp <= 2.0:
        print("WARNING: Exponent for the vehicle dynamics model should be in the range [1.0, 2.0]!")
    if v_max is None:
        v_max = float(np.amin(ggv[-1, 0]))
    ggv = ggv[ggv[:, 0] <= max(v_max, ggv[0, 0]) + 1e-9] if ggv.shape[0] > 1 else ggv
    radii = np.abs(np.divide(1.0, kappa, out=np.full(kappa.size, np.inf), where=kappa != 0.0))

    vx = np.sqrt(np.amin(ggv[:, 2]) * radii)
    for _ in range(100):
        vx_new = np.sqrt(np.interp(np.minimum(vx, v_max), ggv[:, 0], ggv[:, 2]) * radii)
        done = np.nanmax(np.abs(np.where(np.isfinite(vx_new), vx / np.where(vx_new > 0, vx_new, 1.0) - 1.0, 0.0))) < 0.005
        vx = vx_new
        if done:
            break
    vx = np.minimum(vx, v_max)

    n = vx.size
    vx2 = np.concatenate((vx, vx))
    rad2 = np.concatenate((radii, radii))
    el2 = np.concatenate((el_lengths, el_lengths))
    vx2 = _sweep(vx2, rad2, el2, ggv, ax_max_machines, "accel", dyn_model_exp, drag_coeff, m_veh)
    back = _sweep(vx2[::-1], rad2[::-1], np.roll(el2, 1)[::-1] if False else np.concatenate((el2[-1:], el2[:-1]))[::-1],
                  ggv, ax_max_machines, "decel", dyn_model_exp, drag_coeff, m_veh)
    vx2 = back[::-1]
    out = vx2[n:]
    if filt_window is not None:
        out = _cf.conv_filt(signal=out, filt_window=filt_window, closed=True)
    return out
