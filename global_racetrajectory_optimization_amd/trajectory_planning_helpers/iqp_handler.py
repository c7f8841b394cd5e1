"""
Drop-in for tph.iqp_handler.iqp_handler -- boundary [REF main_globaltraj.py:273-284].

Iterated re-linearisation: each pass is one minimum-curvature QP on the MI355X engine,
followed by the raceline re-sampling glue (create_raceline / interp_track_widths / re-spline).  N changes from pass to pass,
so a batch of IQP runs is driven in lock-step rounds with ragged N: every round is ONE batched QP launch over the tracks that
have not terminated yet.  The default is the engine's own loop (mcq_iqp_batch: glue, termination test and damping on the
device, the host reads one int per round); the host-glue driver below is the upstream chain written out, kept as its check.
"""
import time

import numpy as np

from .. import engine as _engine
from . import create_raceline as _cr
from . import interp_track_widths as _itw
from . import opt_min_curv as _omc
from .calc_splines import closed_spline_coeffs as _closed_spline_coeffs


def _respline_normals(refline):
    """normals of calc_splines(use_dist_scaling=False) without building the dense 4N x 4N matrix."""
    n = refline.shape[0]
    _, b, _, _ = _closed_spline_coeffs(refline, np.ones(n))
    nv = np.stack((b[:, 1], -b[:, 0]), axis=1)
    return nv / np.sqrt(np.sum(nv ** 2, axis=1))[:, None]


def _relinearise(reftrack_tmp, normvec_tmp, alpha, stepsize_interp):
    refline_tmp, _, _, _, spline_inds, t_values = _cr.create_raceline(
        refline=reftrack_tmp[:, :2], normvectors=normvec_tmp, alpha=alpha, stepsize_interp=stepsize_interp)[:6]
    reftrack_tmp[:, 2] -= alpha
    reftrack_tmp[:, 3] += alpha
    ws = _itw.interp_track_widths(w_track=reftrack_tmp[:, 2:], spline_inds=spline_inds, t_values=t_values,
                                  incl_last_point=False)
    reftrack_new = np.column_stack((refline_tmp, ws))
    return reftrack_new, _respline_normals(reftrack_new[:, :2])


def _iqp_batch_device(eng, tracks, kappa_bound, w_veh, stepsize_interp, iters_min, curv_error_allowed, print_debug,
                      max_rounds, stats, warm_start=True):
    """The whole batch of IQP runs as ONE engine call (mcq_iqp_batch): QP passes, termination test, damping and the glue
    between the passes all run on the device; the host packs the tracks, waits, and unpacks the
    end states.  The per-round curvature errors come back as a trace, for the print_debug lines upstream prints per round."""
    t_start = time.perf_counter()
    if print_debug:
        # upstream prints one line per iteration AS IT GOES [REF main_globaltraj.py:270,280]: the engine calls back after every QP pass
        # (mcq_iqp_set_round_callback), however many rounds there are
        def on_round(rnd, curv, live):
            for k in np.nonzero(live)[0]:
                print("Minimum curvature IQP: iteration %i, curv_error_max: %.4frad/m" % (rnd, curv[k]), flush=True)
        eng.set_iqp_round_callback(on_round)
    try:
        out = eng.iqp_batch(tracks, kappa_bound, w_veh, stepsize_interp, iters_min, curv_error_allowed, max_rounds,
                            timed=bool(stats is not None and stats.get("timed")), warm_start=0 if warm_start else -1)
    finally:
        if print_debug:
            eng.set_iqp_round_callback(None)
    for k in range(len(tracks)):
        if int(out["status"][k]) == _engine.STATUS_ITER_CAP and int(out["rounds"][k]) >= max_rounds:
            raise RuntimeError("iqp_handler: no convergence within %d rounds" % max_rounds)
        if int(out["status"][k]) == _engine.STATUS_RING_OVERFLOW:
            raise RuntimeError("iqp_handler: re-sampled raceline of track %d does not fit the device buffers "
                               "(nmax = %d)" % (k, out["stats"]["nmax"]))
        _omc.raise_for_status(int(out["status"][k]))
        if print_debug:
            print("Finished IQP!")
    if stats is not None:
        stats.update(out["stats"], device_resident=True, seconds_total=time.perf_counter() - t_start)
    return [(out["alpha"][k], out["reftrack"][k], out["normvectors"][k]) for k in range(len(tracks))]


def iqp_handler_batch(tracks: list, kappa_bound: float, w_veh: float, stepsize_interp: float, iters_min: int = 3,
                      curv_error_allowed: float = 0.01, print_debug: bool = False, engine=None, max_rounds: int = 50,
                      stats: dict = None, device_resident: bool = True, warm_start: bool = True) -> list:
    """tracks: list of dicts {reftrack [N,4], normvectors [N,2], scaling [N] or None}.

    Returns a list of (alpha, reftrack, normvectors) like iqp_handler.  `stats` (optional dict) receives
    {'rounds', 'qp_solves', ...}.  device_resident=True (the default, and what the drop-in iqp_handler runs) is ONE engine
    call: the tracks stay in HBM between the passes, the glue runs as a HIP kernel, termination and damping are decided on
    the device and -- warm_start -- passes 2+ start the exchange from the previous pass's working set instead of the interior
    point (same vertex, a fraction of the factorisations).  device_resident=False runs the glue on the host exactly as
    upstream chains it, one engine launch per round (kept as the cross-check of the device chain).
    """
    eng = engine or _engine.default_engine()
    if device_resident:
        return _iqp_batch_device(eng, tracks, kappa_bound, w_veh, stepsize_interp, iters_min, curv_error_allowed,
                                 print_debug, max_rounds, stats, warm_start)
    state = [dict(ref=np.array(t["reftrack"], dtype=np.float64), nv=np.array(t["normvectors"], dtype=np.float64),
                  sc=None if t.get("scaling") is None else np.array(t["scaling"], dtype=np.float64), done=False,
                  alpha=None) for t in tracks]
    n_solves = 0
    it = 0
    while True:
        it += 1
        live = [k for k, s in enumerate(state) if not s["done"]]
        if not live:
            break
        if it > max_rounds:
            raise RuntimeError("iqp_handler: no convergence within %d rounds" % max_rounds)
        probs = [dict(reftrack=state[k]["ref"], normvec=state[k]["nv"], scaling=state[k]["sc"], kappa_bound=kappa_bound,
                      w_veh=w_veh) for k in live]
        alphas, curv, status, _ = eng.solve_batch(probs)
        n_solves += len(live)
        for j, k in enumerate(live):
            _omc.raise_for_status(int(status[j]))
            s = state[k]
            alpha = alphas[j]
            if print_debug:
                print("Minimum curvature IQP: iteration %i, curv_error_max: %.4frad/m" % (it, curv[j]))
            if it < iters_min:
                alpha = alpha * (it * 1.0 / iters_min)
            s["alpha"] = alpha
            if it >= iters_min and curv[j] <= curv_error_allowed:
                if print_debug:
                    print("Finished IQP!")
                s["done"] = True
                continue
            s["ref"], s["nv"] = _relinearise(s["ref"], s["nv"], alpha, stepsize_interp)
            s["sc"] = None      # re-spline uses use_dist_scaling=False  
    if stats is not None:
        stats.update(rounds=it - 1, qp_solves=n_solves)
    return [(s["alpha"], s["ref"], s["nv"]) for s in state]


def iqp_handler(reftrack: np.ndarray, normvectors: np.ndarray, A: np.ndarray, kappa_bound: float, w_veh: float,
                print_debug: bool, plot_debug: bool, stepsize_interp: float, iters_min: int = 3,
                curv_error_allowed: float = 0.01) -> tuple:
    """Returns (alpha_mincurv [N'], reftrack [N',4], normvectors [N',2]) of the last re-linearisation."""
    reftrack = np.asarray(reftrack, dtype=np.float64)
    normvectors = np.asarray(normvectors, dtype=np.float64)
    _omc._validate(reftrack, normvectors, A, True)
    sc = _engine.les_scalings(A) if A is not None else None
    # upstream aliases the caller's reftrack on the first pass and mutates its width columns in place; main rebinds the
    # name [REF main_globaltraj.py:274], so working on a copy is unobservable there and safer for other callers.
    return iqp_handler_batch([dict(reftrack=reftrack, normvectors=normvectors, scaling=sc)], kappa_bound, w_veh,
                             stepsize_interp, iters_min, curv_error_allowed, print_debug)[0]
