"""
Drop-in for tph.iqp_handler.iqp_handler -- boundary [REF main_globaltraj.py:273-284].

Iterated re-linearisation: each pass is one minimum-curvature QP on the MI355X engine,
followed by the raceline re-sampling glue (create_raceline / interp_track_widths / re-spline) on the host.  N changes
from pass to pass, so a batch of IQP runs is driven in lock-step rounds with ragged N (iqp_handler_batch): every
round is ONE batched engine launch over the tracks that have not terminated yet.
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
    """The lock-step IQP rounds with everything but a few scalars per track resident in HBM: the QP pass is
    mcq_solve_device_ragged, the glue between passes mcq_relinearise_device.  Per round the host
    reads back curv_error / status / N per track and, for the tracks that finish, their final alpha / reftrack /
    normals.  Device memory through the engine's own C ABI (no second HIP runtime in the process)."""
    t_start = time.perf_counter()
    t_down = 0.0
    bsz = len(tracks)
    refs = [np.ascontiguousarray(t["reftrack"], dtype=np.float64) for t in tracks]
    nvs = [np.ascontiguousarray(t["normvectors"], dtype=np.float64) for t in tracks]
    n_host = np.array([r.shape[0] for r in refs], dtype=np.int32)
    # capacity for the re-sampled rings: the raceline is never much longer than the polygon through the reference points;
    # 30 % + 16 points of headroom, reported (not truncated) if it ever is not enough
    nmax = 0
    for r in refs:
        length = float(np.hypot(np.diff(r[:, 0], append=r[0, 0]), np.diff(r[:, 1], append=r[0, 1])).sum())
        nmax = max(nmax, r.shape[0], int(np.ceil(1.3 * length / stepsize_interp)) + 16)
    ref_h = np.zeros((bsz, nmax, 4))
    nv_h = np.zeros((bsz, nmax, 2))
    sc_h = np.ones((bsz, nmax))
    for k in range(bsz):
        ref_h[k, :n_host[k]] = refs[k]
        nv_h[k, :n_host[k]] = nvs[k]
        if tracks[k].get("scaling") is not None:
            sc_h[k, :n_host[k]] = tracks[k]["scaling"]
    f8, i4 = 8, 4
    bufs = []

    def dalloc(nbytes, init=None):
        p = eng.alloc(nbytes)
        bufs.append(p)
        if init is not None:
            eng.upload(p, init)
        return p

    try:
        d_ref = [dalloc(ref_h.nbytes, ref_h), dalloc(ref_h.nbytes)]
        d_nv = [dalloc(nv_h.nbytes, nv_h), dalloc(nv_h.nbytes)]
        d_sc = dalloc(sc_h.nbytes, sc_h)
        d_alpha = dalloc(bsz * nmax * f8)
        d_curv = dalloc(bsz * f8)
        d_status = dalloc(bsz * i4)
        d_n = [dalloc(bsz * i4, n_host), dalloc(bsz * i4)]
        d_nsolve = dalloc(bsz * i4)
        d_live = dalloc(bsz * i4)
        d_rst = dalloc(bsz * i4)
        live = np.ones(bsz, dtype=bool)
        out = [None] * bsz
        cur, n_solves, it = 0, 0, 0
        eng.sync()
        t_up = time.perf_counter() - t_start       # marshalling + upload of the tracks
        while live.any():
            it += 1
            if it > max_rounds:
                raise RuntimeError("iqp_handler: no convergence within %d rounds" % max_rounds)
            # finished tracks keep their buffers but are skipped: n = 0 makes the assembly kernel flag them, the solver returns
            eng.upload(d_nsolve, (n_host * live).astype(np.int32))
            # passes 2+: the exchange starts from the working set of the previous pass, which the glue kernel carried over
            eng.solve_device_ragged(bsz, nmax, d_nsolve, d_ref[cur], d_nv[cur], d_sc if it == 1 else None, kappa_bound,
                                    w_veh, d_alpha, d_curv, d_status, warm_start=1 if (warm_start and it > 1) else 0)
            n_solves += int(live.sum())
            curv = eng.download(d_curv, (bsz,), np.float64)
            status = eng.download(d_status, (bsz,), np.int32)
            scale = it * 1.0 / iters_min if it < iters_min else 1.0
            t_d0 = time.perf_counter()
            done = []
            for k in np.nonzero(live)[0]:
                _omc.raise_for_status(int(status[k]))
                if print_debug:
                    print("Minimum curvature IQP: iteration %i, curv_error_max: %.4frad/m" % (it, curv[k]))
                if it >= iters_min and curv[k] <= curv_error_allowed:
                    if print_debug:
                        print("Finished IQP!")
                    done.append(int(k))
            if len(done) > 8:
                # many tracks finish in the same round (the usual case: identical iters_min): three bulk copies instead of
                # three small blocking copies per track
                al_all = eng.download(d_alpha, (bsz, nmax), np.float64)
                ref_all = eng.download(d_ref[cur], (bsz, nmax, 4), np.float64)
                nv_all = eng.download(d_nv[cur], (bsz, nmax, 2), np.float64)
                for k in done:          # views into the three bulk arrays (no per-track copies)
                    nk = int(n_host[k])
                    out[k] = (al_all[k, :nk], ref_all[k, :nk], nv_all[k, :nk])
            else:
                for k in done:
                    nk = int(n_host[k])
                    out[k] = (eng.download(d_alpha, (nk,), np.float64, k * nmax * f8),
                              eng.download(d_ref[cur], (nk, 4), np.float64, k * nmax * 4 * f8),
                              eng.download(d_nv[cur], (nk, 2), np.float64, k * nmax * 2 * f8))
            live[done] = False
            t_down += time.perf_counter() - t_d0       # read-back of the tracks that finished in this round
            if not live.any():
                break
            eng.upload(d_live, live.astype(np.int32))
            eng.relinearise_device(bsz, nmax, d_n[cur], d_ref[cur], d_nv[cur], d_alpha, d_live, scale, stepsize_interp,
                                   d_ref[1 - cur], d_nv[1 - cur], d_n[1 - cur], d_rst)
            rst = eng.download(d_rst, (bsz,), np.int32)
            n_new = eng.download(d_n[1 - cur], (bsz,), np.int32)
            for k in np.nonzero(live)[0]:
                if rst[k] != 0:
                    raise RuntimeError("iqp_handler: re-sampled raceline of track %d does not fit the device buffers "
                                       "(nmax = %d)" % (k, nmax))
                n_host[k] = n_new[k]
            cur = 1 - cur
    finally:
        for p in bufs:
            eng.free(p)
    if stats is not None:
        total = time.perf_counter() - t_start
        stats.update(rounds=it, qp_solves=n_solves, device_resident=True, nmax=nmax, seconds_upload=t_up,
                     seconds_download=t_down, seconds_rounds=total - t_up - t_down)
    return out


def iqp_handler_batch(tracks: list, kappa_bound: float, w_veh: float, stepsize_interp: float, iters_min: int = 3,
                      curv_error_allowed: float = 0.01, print_debug: bool = False, engine=None, max_rounds: int = 50,
                      stats: dict = None, device_resident: bool = False, warm_start: bool = True) -> list:
    """tracks: list of dicts {reftrack [N,4], normvectors [N,2], scaling [N] or None}.

    Returns a list of (alpha, reftrack, normvectors) like iqp_handler.  `stats` (optional dict) receives
    {'rounds', 'qp_solves'}.  device_resident=True keeps the tracks in HBM between the passes (the glue runs as a HIP
    kernel, mcq_relinearise_device, and -- warm_start -- passes 2+ start the exchange from the previous pass's working set
    instead of the interior point: same vertex, a fraction of the factorisations); the default runs the glue on the host
    exactly as upstream chains it.
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
    from .calc_splines import scalings_from_les_matrix
    reftrack = np.asarray(reftrack, dtype=np.float64)
    normvectors = np.asarray(normvectors, dtype=np.float64)
    _omc._validate(reftrack, normvectors, A, True)
    sc = scalings_from_les_matrix(A) if A is not None else None
    # upstream aliases the caller's reftrack on the first pass and mutates its width columns in place; main rebinds the
    # name [REF main_globaltraj.py:274], so working on a copy is unobservable there and safer for other callers.
    return iqp_handler_batch([dict(reftrack=reftrack, normvectors=normvectors, scaling=sc)], kappa_bound, w_veh,
                             stepsize_interp, iters_min, curv_error_allowed, print_debug)[0]
