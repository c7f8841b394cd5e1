"""
Drop-in for tph.opt_min_curv.opt_min_curv -- boundary [REF main_globaltraj.py:264-271, 344-350].

Same signature, argument meaning, return value and exception types as the third-party function the reference calls
(SURVEY.md section 8b / App. A).  The work is done by the MI355X engine (hand-written HIP behind the C ABI of
include/mcq.h); this module only validates, extracts the N spline scalings from the dense matrix `A`, and maps the
engine's per-problem status to the exceptions tph/quadprog would raise.  There is no CPU solve here: if libmcq.so is
not available the call raises EngineError.
"""
import time

import numpy as np

from .. import engine as _engine

_MSG_TOO_SMALL = "Problem not solvable, track might be too small to run with current safety distance!"


def _raise_for_status(status: int) -> None:
    if status == _engine.STATUS_OK:
        return
    if status == _engine.STATUS_INFEASIBLE:
        raise RuntimeError(_MSG_TOO_SMALL)
    if status == _engine.STATUS_NOT_PD:
        raise ValueError("matrix G is not positive definite")
    if status == _engine.STATUS_KAPPA_INFEASIBLE:
        raise ValueError("constraints are inconsistent, no solution")
    if status == _engine.STATUS_KAPPA_ACTIVE:
        raise RuntimeError("opt_min_curv (MI355X engine): a curvature-bound row is violated at the returned point (status 6: the "
                           "curvature rows were switched off with check_kappa < 0, or the solution missed the bound by more than 1e-7)")
    if status == _engine.STATUS_BAD_INPUT:
        raise RuntimeError("opt_min_curv (MI355X engine): non-finite input or fewer than 3 points (status 4)")
    if status == _engine.STATUS_RING_OVERFLOW:
        raise RuntimeError("opt_min_curv (MI355X engine): the re-sampled raceline outgrew the caller's buffers (status 7)")
    if status == _engine.STATUS_KAPPA_NO_SLOT:          # (not returned since round 5: the Goldfarb-Idnani path takes such problems)
        raise RuntimeError("opt_min_curv (MI355X engine): no overflow slot for the curvature-row working set in this launch (status 8): "
                           "solve the problem again in a launch of its own")
    raise RuntimeError("opt_min_curv (MI355X engine): iteration cap reached (status %d) -- the block-pivoting phase AND the Goldfarb-Idnani "
                       "fallback ran out of their budgets" % status)


def _validate(reftrack, normvectors, A, closed):
    no_points = reftrack.shape[0]
    if not closed:
        raise NotImplementedError("MI355X opt_min_curv: only closed tracks are supported (the reference never passes "
                                  "closed=False, SURVEY.md section 8b)")
    if no_points != normvectors.shape[0]:
        raise RuntimeError("Array size of reftrack should be the same as normvectors!")
    if A is not None and (no_points * 4 != A.shape[0] or A.shape[0] != A.shape[1]):
        raise RuntimeError("Spline equation system matrix A has wrong dimensions!")


def opt_min_curv(reftrack: np.ndarray, normvectors: np.ndarray, A: np.ndarray, kappa_bound: float, w_veh: float,
                 print_debug: bool = False, plot_debug: bool = False, closed: bool = True, psi_s: float = None,
                 psi_e: float = None, fix_s: bool = False, fix_e: bool = False) -> tuple:
    """Returns (alpha_mincurv [N], curv_error_max) -- main_globaltraj.py keeps [0]."""
    reftrack = np.asarray(reftrack, dtype=np.float64)
    normvectors = np.asarray(normvectors, dtype=np.float64)
    _validate(reftrack, normvectors, A, closed)
    scaling = _engine.les_scalings(A) if A is not None else None      # (threaded C pass over the dense matrix: engine.les_scalings)

    eng = _engine.default_engine()
    t_start = time.perf_counter()
    alphas, curv, status, _ = eng.solve_batch([dict(reftrack=reftrack, normvec=normvectors, scaling=scaling,
                                                    kappa_bound=kappa_bound, w_veh=w_veh)])
    if print_debug:
        print("Solver runtime opt_min_curv: " + "{:.3f}".format(time.perf_counter() - t_start) + "s")
    _raise_for_status(int(status[0]))
    return alphas[0], float(curv[0])


def opt_min_curv_batch(problems: list, engine=None, **opt_kw) -> tuple:
    """Batch axis of the engine exposed with the same per-problem contract.

    problems: list of dicts {reftrack, normvectors, scaling (or A), kappa_bound, w_veh}.  Independent tracks, vehicle-
    width sweeps and IQP re-linearisations all go through here.  Returns (alphas, curv_errs,
    status, infos); callers decide how to treat non-zero status (raise_for_status mirrors the single-problem errors).
    """
    eng = engine or _engine.default_engine()
    packed = []
    for p in problems:
        ref = np.asarray(p["reftrack"], dtype=np.float64)
        nv = np.asarray(p["normvectors"], dtype=np.float64)
        _validate(ref, nv, p.get("A"), True)
        sc = p.get("scaling")
        if sc is None and p.get("A") is not None:
            sc = _engine.les_scalings(p["A"])
        packed.append(dict(reftrack=ref, normvec=nv, scaling=sc, kappa_bound=p["kappa_bound"], w_veh=p["w_veh"]))
    return eng.solve_batch(packed, **opt_kw)


raise_for_status = _raise_for_status
