"""
ORACLE -- TEST INFRASTRUCTURE ONLY (header of oracle/tph_ref.py applies: only tests/, smoke() and bench.py's cpu_baseline leg
may import this; the product never does).  PARITY UNPINNED BY THE REFERENCE.

CPU restatement of the helpers either side of the QP that SURVEY.md section 8 rows f-2 / f-3 put on the device:

  calc_vel_profile (+ calc_ax_poss)   <- [REF main_globaltraj.py:400-410, 469-479]
  calc_ax_profile                     <- [REF main_globaltraj.py:413-416, 482-485]
  calc_t_profile                      <- [REF main_globaltraj.py:419-421, 488-490]
  check_normals_crossing              <- [REF helper_funcs_glob/src/prep_track.py:57-59]

All four live in the third-party package trajectory_planning_helpers==0.76 [REF requirements.txt:3], which is neither
vendored under /root/reference nor installable here; the reference holds no tests or expected outputs for them.  This file
restates the published algorithm of that package function by function, INCLUDING the details that change numbers:

  * calc_vel_profile: the lateral limit is a fixed-point iteration on numpy.interp over the ggv rows (all rows: nothing is
    truncated at v_max; the function raises if ggv / ax_max_machines end below v_max), stopped when the largest relative
    change is below 0.5 % (a NaN in that maximum -- kappa == 0 gives inf / inf -- never satisfies the test, upstream then
    runs all 100 rounds);
  * the forward (acceleration) sweep runs over the lap doubled, the backward (deceleration) sweep over the SECOND lap of
    the forward result doubled again, the second lap of that is returned;
  * a sweep starts only at the first point of every acceleration phase of the profile it is handed (v[i+1] > v[i] and not
    v[i] > v[i-1]), runs on through later phase starts, and stops where the attainable speed exceeds v_max until the next
    phase start;
  * the backward sweep recomputes the available deceleration ONE step ahead (at the attainable speed, with the radius of
    the next point) and keeps the smaller speed ("looping just once at the moment");
  * arrays are flipped as a whole for the backward sweep, so step i of it uses the element length stored at the point it
    leaves (el_lengths_mod[i]), as forward.

Python loops over the points: use on a few variants of one track, not on batches.
"""
import math

import numpy as np


def calc_ax_poss(vx_start, radius, ggv, mu, dyn_model_exp, drag_coeff, m_veh, ax_max_machines=None, mode="accel_forw", loc_gg=None):
    """Longitudinal acceleration still available at vx_start on `radius` (tyre potential shared with the lateral
    acceleration through the generalised friction ellipse, machine limit when accelerating, drag).  Tyre potential from the
    speed-dependent ggv diagram, or (round 6: the `loc_gg` form of tph.calc_vel_profile, which the reference's mintime branch with a
    variable friction map would reach [REF main_globaltraj.py:396-410]) from the local pair loc_gg = (ax_max, ay_max) of the point."""
    if mode not in ("accel_forw", "decel_forw", "decel_backw"):
        raise RuntimeError("Unknown operation mode for calc_ax_poss!")
    if mode == "accel_forw" and ax_max_machines is None:
        raise RuntimeError("ax_max_machines is required if operation mode is accel_forw!")
    if ggv is not None and (ggv.ndim != 2 or ggv.shape[1] != 3):
        raise RuntimeError("ggv must have two dimensions and three columns [vx, ax_max, ay_max]!")
    if ggv is not None:
        ax_max_tires = mu * np.interp(vx_start, ggv[:, 0], ggv[:, 1])
        ay_max_tires = mu * np.interp(vx_start, ggv[:, 0], ggv[:, 2])
    else:
        ax_max_tires = mu * loc_gg[0]
        ay_max_tires = mu * loc_gg[1]
    ay_used = math.pow(vx_start, 2) / radius
    if mode in ("accel_forw", "decel_backw") and ax_max_tires < 0.0:
        ax_max_tires *= -1.0
    elif mode == "decel_forw" and ax_max_tires > 0.0:
        ax_max_tires *= -1.0
    radicand = 1.0 - math.pow(ay_used / ay_max_tires, dyn_model_exp)
    ax_avail_tires = ax_max_tires * math.pow(radicand, 1.0 / dyn_model_exp) if radicand > 0.0 else 0.0
    if mode == "accel_forw":
        ax_avail_vehicle = min(ax_avail_tires, np.interp(vx_start, ax_max_machines[:, 0], ax_max_machines[:, 1]))
    else:
        ax_avail_vehicle = ax_avail_tires
    ax_drag = -math.pow(vx_start, 2) * drag_coeff / m_veh
    if mode in ("accel_forw", "decel_forw"):
        return ax_avail_vehicle + ax_drag
    return ax_avail_vehicle - ax_drag


def _solver_fb_acc_profile(ggv, ax_max_machines, v_max, radii, el_lengths, mu, vx_profile, dyn_model_exp, drag_coeff, m_veh,
                           backwards=False, loc_gg=None):
    no_points = vx_profile.size
    if backwards:
        radii_mod, el_lengths_mod, mu_mod = np.flipud(radii), np.flipud(el_lengths), np.flipud(mu)
        vx_profile = np.flipud(vx_profile).copy()
        mode = "decel_backw"
    else:
        radii_mod, el_lengths_mod, mu_mod = radii, el_lengths, mu
        vx_profile = vx_profile.copy()
        mode = "accel_forw"
    loc_gg_mod = None if loc_gg is None else (np.flipud(loc_gg) if backwards else loc_gg)
    lg = (lambda j: None) if loc_gg_mod is None else (lambda j: loc_gg_mod[j])
    # start points of the acceleration phases of the profile as handed in
    acc_inds = np.where(np.diff(vx_profile) > 0.0)[0]
    if acc_inds.size != 0:
        acc_inds_diffs = np.insert(np.diff(acc_inds), 0, 2)
        acc_inds_rel = list(acc_inds[acc_inds_diffs > 1])
    else:
        acc_inds_rel = []
    while acc_inds_rel:
        i = acc_inds_rel.pop(0)
        while i < no_points - 1:
            ax_possible_cur = calc_ax_poss(vx_profile[i], radii_mod[i], ggv, mu_mod[i], dyn_model_exp, drag_coeff, m_veh,
                                           ax_max_machines, mode, loc_gg=lg(i))
            vx_possible_next = math.sqrt(math.pow(vx_profile[i], 2) + 2 * ax_possible_cur * el_lengths_mod[i])
            if backwards:
                # the deceleration available at point i need not be available at the next one: one look-ahead round
                for _ in range(1):
                    ax_possible_next = calc_ax_poss(vx_possible_next, radii_mod[i + 1], ggv, mu_mod[i + 1], dyn_model_exp,
                                                    drag_coeff, m_veh, ax_max_machines, mode, loc_gg=lg(i + 1))
                    vx_tmp = math.sqrt(math.pow(vx_profile[i], 2) + 2 * ax_possible_next * el_lengths_mod[i])
                    if vx_tmp < vx_possible_next:
                        vx_possible_next = vx_tmp
                    else:
                        break
            if vx_possible_next < vx_profile[i + 1]:
                vx_profile[i + 1] = vx_possible_next
            i += 1
            if vx_possible_next > v_max or (acc_inds_rel and i >= acc_inds_rel[0]):
                break
    return np.flipud(vx_profile) if backwards else vx_profile


def conv_filt(signal, filt_window, closed):
    """tph.conv_filt: moving average, window odd; closed signals wrap around."""
    if filt_window % 2 == 0:
        raise RuntimeError("Window width of moving average filter must be odd!")
    w = int((filt_window - 1) / 2)
    if closed:
        tmp = np.concatenate((signal[-w:], signal, signal[:w]), axis=0)
        return np.convolve(tmp, np.ones(filt_window) / float(filt_window), mode="same")[w:-w]
    out = np.copy(signal)
    out[w:-w] = np.convolve(signal, np.ones(filt_window) / float(filt_window), mode="same")[w:-w]
    return out


def calc_vel_profile(ax_max_machines, kappa, el_lengths, closed, drag_coeff, m_veh, ggv=None, loc_gg=None, v_max=None,
                     dyn_model_exp=1.0, mu=None, v_start=None, v_end=None, filt_window=None, info=None):
    """Closed tracks with a global ggv (optionally mu) are the form every call site of the reference's in-scope flow uses; round 6 adds the two
    other forms of upstream's signature -- `loc_gg` [no_points, 2] = local (ax_max, ay_max) per point instead of the ggv diagram (v_max is then
    mandatory), and unclosed profiles (kappa one longer than el_lengths; v_start mandatory, v_end optional: the sweeps run once over the profile
    instead of over the lap doubled).  `info` (dict) receives the number of fixed-point rounds of the lateral limit."""
    if (ggv is not None or mu is not None) and loc_gg is not None:
        raise RuntimeError("Either ggv and optionally mu OR loc_gg must be supplied, not both (or all) of them!")
    if ggv is None and loc_gg is None:
        raise RuntimeError("Either ggv or loc_gg must be supplied!")
    if loc_gg is not None:
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
        v_start = 0.0
    if v_end is not None and v_end < 0.0:
        v_end = 0.0
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
    if mu is None:
        mu = np.ones(kappa.size)
    no_points = radii.size

    # lateral limit
    rounds = 0
    if ggv is not None:
        mu_mean = float(np.mean(mu))                       # upstream: the first estimate uses the mean friction coefficient
        ay_max_global = mu_mean * np.amin(ggv[:, 2])
        vx_profile = np.sqrt(ay_max_global * radii)
        for _ in range(100):
            rounds += 1
            vx_prev = vx_profile
            ay_max_curr = mu * np.interp(vx_profile, ggv[:, 0], ggv[:, 2])
            vx_profile = np.sqrt(np.multiply(ay_max_curr, radii))
            with np.errstate(invalid="ignore", divide="ignore"):
                if np.max(np.abs(vx_profile / vx_prev - 1.0)) < 0.005:
                    break
    else:
        vx_profile = np.sqrt(loc_gg[:, 1] * radii)         # the local lateral limit does not depend on the speed: no iteration
    if info is not None:
        info["lateral_rounds"] = rounds
    vx_profile[vx_profile > v_max] = v_max

    if not closed:
        # once over the profile: the start speed caps the first point, the end speed (if given) the last one
        if vx_profile[0] > v_start:
            vx_profile[0] = v_start
        vx_profile = _solver_fb_acc_profile(ggv, ax_max_machines, v_max, radii, el_lengths, mu, vx_profile, dyn_model_exp, drag_coeff, m_veh,
                                            backwards=False, loc_gg=loc_gg)
        if v_end is not None and vx_profile[-1] > v_end:
            vx_profile[-1] = v_end
        vx_profile = _solver_fb_acc_profile(ggv, ax_max_machines, v_max, radii, el_lengths, mu, vx_profile, dyn_model_exp, drag_coeff, m_veh,
                                            backwards=True, loc_gg=loc_gg)
    else:
        # forward over two laps, backward over the doubled second lap of that
        radii_d = np.concatenate((radii, radii))
        el_d = np.concatenate((el_lengths, el_lengths))
        mu_d = np.concatenate((mu, mu))
        lg_d = None if loc_gg is None else np.concatenate((loc_gg, loc_gg), axis=0)
        vx_d = _solver_fb_acc_profile(ggv, ax_max_machines, v_max, radii_d, el_d, mu_d, np.concatenate((vx_profile, vx_profile)),
                                      dyn_model_exp, drag_coeff, m_veh, backwards=False, loc_gg=lg_d)
        vx_d = np.concatenate((vx_d[no_points:], vx_d[no_points:]))
        vx_d = _solver_fb_acc_profile(ggv, ax_max_machines, v_max, radii_d, el_d, mu_d, vx_d, dyn_model_exp, drag_coeff, m_veh,
                                      backwards=True, loc_gg=lg_d)
        vx_profile = vx_d[no_points:]
    if filt_window is not None:
        vx_profile = conv_filt(vx_profile, filt_window, closed)
    return vx_profile


def calc_ax_profile(vx_profile, el_lengths, eq_length_output=False):
    if vx_profile.size != el_lengths.size + 1:
        raise RuntimeError("Array size of vx_profile should be 1 element bigger than el_lengths!")
    ax = (np.power(vx_profile[1:], 2) - np.power(vx_profile[:-1], 2)) / (2 * el_lengths)
    if eq_length_output:
        out = np.zeros(vx_profile.size)
        out[:-1] = ax
        return out
    return ax


def calc_t_profile(vx_profile, el_lengths, t_start=0.0, ax_profile=None):
    if vx_profile.size < el_lengths.size:
        raise RuntimeError("vx_profile and el_lenghts must have at least the same length!")
    if ax_profile is not None and ax_profile.size < el_lengths.size:
        raise RuntimeError("ax_profile and el_lenghts must have at least the same length!")
    if ax_profile is None:
        ax_profile = calc_ax_profile(vx_profile, el_lengths, False)
    no_points = el_lengths.size
    t_steps = np.zeros(no_points)
    for i in range(no_points):
        if not math.isclose(ax_profile[i], 0.0):
            t_steps[i] = (-vx_profile[i] + math.sqrt(math.pow(vx_profile[i], 2) + 2 * ax_profile[i] * el_lengths[i])) \
                / ax_profile[i]
        else:
            t_steps[i] = el_lengths[i] / vx_profile[i]
    return np.insert(np.cumsum(t_steps), 0, 0.0) + t_start


def lap_time_stable(vx_profile, el_lengths):
    """Lap time of a closed profile as the sum of 2 l / (v_a + v_b): algebraically calc_t_profile's per-element expression
    (constant acceleration over the element), without its cancellation as the acceleration goes to zero -- upstream's form
    turns a 1e-15 ripple of v on a constant-speed stretch into up to seconds per element, so parity of lap times is stated
    on this form (the device kernel computes it this way) and only loosely on calc_t_profile's own last entry."""
    v_cl = np.append(vx_profile, vx_profile[0])
    return float(np.sum(2.0 * el_lengths / (v_cl[:-1] + v_cl[1:])))


def check_normals_crossing(track, normvec_normalized, horizon=10):
    """Do the normal segments [p - w_left n, p + w_right n] of two points at most `horizon` apart intersect?"""
    no_points = track.shape[0]
    if horizon >= no_points:
        raise RuntimeError("Horizon of %i points is too large for a track with %i points, reduce horizon!"
                           % (horizon, no_points))
    les_mat = np.zeros((2, 2))
    idx_list = list(range(0, no_points))
    idx_list = idx_list[-horizon:] + idx_list + idx_list[:horizon]
    for idx in range(no_points):
        idx_neighbours = idx_list[idx:idx + 2 * horizon + 1]
        del idx_neighbours[horizon]
        idx_neighbours = np.array(idx_neighbours)
        nv0, nvn = normvec_normalized[idx], normvec_normalized[idx_neighbours]
        is_collinear_b = np.isclose(nv0[0] * nvn[:, 1] - nv0[1] * nvn[:, 0], 0.0)
        idx_neighbours_rel = idx_neighbours[np.nonzero(np.invert(is_collinear_b))[0]]
        for idx_comp in list(idx_neighbours_rel):
            const = track[idx_comp, :2] - track[idx, :2]
            les_mat[:, 0] = normvec_normalized[idx]
            les_mat[:, 1] = -normvec_normalized[idx_comp]
            lambdas = np.linalg.solve(les_mat, const)
            if -track[idx, 3] <= lambdas[0] <= track[idx, 2] and -track[idx_comp, 3] <= lambdas[1] <= track[idx_comp, 2]:
                return True
    return False
