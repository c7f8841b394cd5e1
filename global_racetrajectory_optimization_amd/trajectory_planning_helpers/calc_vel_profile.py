"""
Host-side shim of tph.calc_vel_profile -- boundary [REF main_globaltraj.py:400-410, 469-479].

Forward/backward quasi-steady-state velocity profile over the ggv diagram (the batched device
form is mcq_vel_profile_device, this is the single-profile host form the untouched script calls).  Same numbers as upstream
(checked against the restatement in oracle/vel_ref.py by tests/): lateral limit v = sqrt(ay_max(v) R) by fixed-point
iteration over ALL ggv rows, forward sweep over the lap doubled, backward sweep over the doubled second lap of that, sweeps
gated on the starts of the acceleration phases and on v_max, one look-ahead round in the backward sweep.

The sweeps are written as ONE pass with an `active` flag -- the form the device kernel uses -- instead of upstream's work
list of phase starts: a sweep is switched on at every phase start of the profile it was handed (v[i+1] > v[i] and not
v[i] > v[i-1]) and switched off where the attainable speed exceeds v_max.
"""
import math

import numpy as np

from . import conv_filt as _cf


def _ax_possible(v, radius, ggv, ax_max_machines, accel, dyn_model_exp, drag_over_m, mu, lgg=None):
    if lgg is None:
        ax_tires = abs(mu * np.interp(v, ggv[:, 0], ggv[:, 1]))
        ay_tires = mu * np.interp(v, ggv[:, 0], ggv[:, 2])
    else:               # local (ax_max, ay_max) of the point instead of the speed-dependent diagram
        ax_tires = abs(mu * lgg[0])
        ay_tires = mu * lgg[1]
    radicand = 1.0 - math.pow(v * v / radius / ay_tires, dyn_model_exp)
    ax_avail = ax_tires * math.pow(radicand, 1.0 / dyn_model_exp) if radicand > 0.0 else 0.0
    ax_drag = -v * v * drag_over_m
    if accel:
        return min(ax_avail, np.interp(v, ax_max_machines[:, 0], ax_max_machines[:, 1])) + ax_drag
    return ax_avail - ax_drag        # braking, integrated backwards: drag helps


def _sweep(v_in, radii, el, mu, ggv, ax_max_machines, v_max, accel, dyn_model_exp, drag_over_m, loc_gg=None):
    """One gated pass over v_in (arrays already in sweep order).  Phase starts refer to v_in as handed in."""
    v = v_in.copy()
    m = v.size
    rising = np.diff(v_in) > 0.0
    active = False
    for i in range(m - 1):
        if rising[i] and (i == 0 or not rising[i - 1]):
            active = True
        if not active:
            continue
        v2 = v[i] * v[i]
        v_next = math.sqrt(v2 + 2.0 * _ax_possible(v[i], radii[i], ggv, ax_max_machines, accel, dyn_model_exp, drag_over_m,
                                                   mu[i], None if loc_gg is None else loc_gg[i]) * el[i])
        if not accel:       # the deceleration available here need not be available one point on: look ahead once
            v_tmp = math.sqrt(v2 + 2.0 * _ax_possible(v_next, radii[i + 1], ggv, ax_max_machines, accel, dyn_model_exp,
                                                      drag_over_m, mu[i + 1], None if loc_gg is None else loc_gg[i + 1]) * el[i])
            v_next = min(v_next, v_tmp)
        if v_next < v[i + 1]:
            v[i + 1] = v_next
        if v_next > v_max:
            active = False
    return v


def calc_vel_profile(ax_max_machines: np.ndarray, kappa: np.ndarray, el_lengths: np.ndarray, closed: bool,
                     drag_coeff: float, m_veh: float, ggv: np.ndarray = None, loc_gg: np.ndarray = None,
                     v_max: float = None, dyn_model_exp: float = 1.0, mu: np.ndarray = None, v_start: float = None,
                     v_end: float = None, filt_window: int = None) -> np.ndarray:
    if (ggv is not None or mu is not None) and loc_gg is not None:
        raise RuntimeError("Either ggv and optionally mu OR loc_gg must be supplied, not both (or all) of them!")
    if ggv is None and loc_gg is None:
        raise RuntimeError("Either ggv or loc_gg must be supplied!")
    if loc_gg is not None:
        # (round 6: the local-gg form and unclosed profiles -- upstream's whole signature; the reference's in-scope flow uses neither)
        loc_gg = np.asarray(loc_gg, dtype=np.float64)
        if v_max is None:
            raise RuntimeError("v_max must be supplied if loc_gg is used!")
        if loc_gg.ndim != 2 or loc_gg.shape != (kappa.size, 2):
            raise RuntimeError("loc_gg must have the shape [no_points, 2]!")
    if ggv is not None and ggv.shape[1] != 3:
        raise RuntimeError("ggv diagram must consist of the three columns [vx, ax_max, ay_max]!")
    if mu is not None and kappa.size != mu.size:
        raise RuntimeError("kappa and mu must have the same length!")
    if closed and kappa.size != el_lengths.size:
        raise RuntimeError("kappa and el_lengths must have the same length if closed!")
    if not closed and kappa.size != el_lengths.size + 1:
        raise RuntimeError("kappa must have the length of el_lengths + 1 if unclosed!")
    if not closed and v_start is None:
        raise RuntimeError("v_start must be provided for the unclosed case!")
    if v_start is not None and v_start < 0.0:
        print("WARNING: Input v_start was < 0.0. Using v_start = 0.0 instead!")
        v_start = 0.0
    if v_end is not None and v_end < 0.0:
        print("WARNING: Input v_end was < 0.0. Using v_end = 0.0 instead!")
        v_end = 0.0
    if not 1.0 <= dyn_model_exp <= 2.0:
        print("WARNING: Exponent for the vehicle dynamics model should be in the range [1.0, 2.0]!")
    if ax_max_machines.shape[1] != 2:
        raise RuntimeError("ax_max_machines must consist of the two columns [vx, ax_max_machines]!")
    if v_max is None:
        v_max = min(ggv[-1, 0], ax_max_machines[-1, 0])
    else:
        if ggv is not None and ggv[-1, 0] < v_max:
            raise RuntimeError("ggv has to cover the entire velocity range of the car (i.e. >= v_max)!")
        if ax_max_machines[-1, 0] < v_max:
            raise RuntimeError("ax_max_machines has to cover the entire velocity range of the car (i.e. >= v_max)!")
    radii = np.abs(np.divide(1.0, kappa, out=np.full(kappa.size, np.inf), where=kappa != 0.0))
    mu = np.ones(kappa.size) if mu is None else np.asarray(mu, dtype=np.float64)

    if ggv is not None:
        # first estimate of the lateral limit from the MEAN friction coefficient, as upstream (ay_max_global = mu_mean * min(ay_max)); the
        # fixed point below stops on a 0.5 % relative change, so the start matters at that level when mu is not uniform
        vx = np.sqrt(float(np.mean(mu)) * np.amin(ggv[:, 2]) * radii)
        converged = False
        for _ in range(100):
            vx_new = np.sqrt(mu * np.interp(vx, ggv[:, 0], ggv[:, 2]) * radii)
            with np.errstate(invalid="ignore", divide="ignore"):
                worst = np.max(np.abs(vx_new / vx - 1.0))     # NaN (inf / inf where kappa == 0) never passes, as upstream
            vx = vx_new
            if worst < 0.005:
                converged = True
                break
        if not converged:
            print("The initial vx profile did not converge after 100 iterations, please check radii and ggv!")
    else:
        vx = np.sqrt(loc_gg[:, 1] * radii)             # a local lateral limit does not depend on the speed
    vx = np.minimum(vx, v_max)

    n = vx.size
    dom = drag_coeff / m_veh
    if not closed:
        # once over the profile, the start speed on the first point, the end speed (if any) on the last
        vx = vx.copy()
        vx[0] = min(vx[0], v_start)
        fwd = _sweep(vx, radii, el_lengths, mu, ggv, ax_max_machines, v_max, True, dyn_model_exp, dom, loc_gg)
        if v_end is not None:
            fwd[-1] = min(fwd[-1], v_end)
        lgr = None if loc_gg is None else loc_gg[::-1]
        out = _sweep(fwd[::-1], radii[::-1], el_lengths[::-1], mu[::-1], ggv, ax_max_machines, v_max, False, dyn_model_exp, dom, lgr)[::-1]
    else:
        rad2, el2, mu2 = np.concatenate((radii, radii)), np.concatenate((el_lengths, el_lengths)), np.concatenate((mu, mu))
        lg2 = None if loc_gg is None else np.concatenate((loc_gg, loc_gg), axis=0)
        fwd = _sweep(np.concatenate((vx, vx)), rad2, el2, mu2, ggv, ax_max_machines, v_max, True, dyn_model_exp, dom, lg2)
        lap2 = np.concatenate((fwd[n:], fwd[n:]))
        # backward: every array flipped as a whole (step i uses the element length stored at the point it leaves, as upstream)
        bwd = _sweep(lap2[::-1], rad2[::-1], el2[::-1], mu2[::-1], ggv, ax_max_machines, v_max, False, dyn_model_exp, dom,
                     None if lg2 is None else lg2[::-1])
        out = bwd[::-1][n:]
    if filt_window is not None:
        out = _cf.conv_filt(signal=out, filt_window=filt_window, closed=closed)
    return out
