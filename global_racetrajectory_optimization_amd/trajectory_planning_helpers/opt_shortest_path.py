"""
Drop-in for tph.opt_shortest_path.opt_shortest_path -- boundary [REF main_globaltraj.py:286-290] (SURVEY.md section 8
row f-4).

Same signature, argument meaning and return value as the third-party function the reference calls: the lateral shifts
alpha [N] (positive to the right, along the normal vectors) that minimise the length of the closed polygon through
p_i + alpha_i n_i, with every shift kept inside the track minus half the vehicle width (deviations smaller than 1 mm are
clipped to 1 mm, as upstream does, instead of being rejected).  The QP runs on the MI355X engine (objective
MCQ_OBJ_SHORTEST_PATH of include/mcq.h: the assembly kernel writes the cyclic tridiagonal H and f, the box-QP solver of
the minimum-curvature path does the rest).  No CPU solve: without libmcq.so the call raises EngineError.
"""
import time

import numpy as np

from .. import engine as _engine


def _raise_for_status(status: int) -> None:
    if status == _engine.STATUS_OK:
        return
    if status == _engine.STATUS_NOT_PD:
        raise ValueError("matrix G is not positive definite")
    if status == _engine.STATUS_BAD_INPUT:
        raise RuntimeError("opt_shortest_path (MI355X engine): non-finite input or fewer than 3 points (status 4)")
    raise RuntimeError("opt_shortest_path (MI355X engine): iteration cap reached (status %d)" % status)


def _validate(reftrack, normvectors):
    if reftrack.ndim != 2 or reftrack.shape[1] != 4:
        raise RuntimeError("reftrack must be [x, y, w_tr_right, w_tr_left] rows!")
    if reftrack.shape[0] != normvectors.shape[0]:
        raise RuntimeError("Array size of reftrack should be the same as normvectors!")


def opt_shortest_path(reftrack: np.ndarray, normvectors: np.ndarray, w_veh: float,
                      print_debug: bool = False) -> np.ndarray:
    """Returns alpha_shpath [N]."""
    reftrack = np.asarray(reftrack, dtype=np.float64)
    normvectors = np.asarray(normvectors, dtype=np.float64)
    _validate(reftrack, normvectors)
    eng = _engine.default_engine()
    t_start = time.perf_counter()
    alphas, _, status, _ = eng.solve_batch([dict(reftrack=reftrack, normvec=normvectors, scaling=None, kappa_bound=1.0,
                                                 w_veh=w_veh)], objective=_engine.OBJ_SHORTEST_PATH)
    if print_debug:
        print("Solver runtime opt_shortest_path: " + "{:.3f}".format(time.perf_counter() - t_start) + "s")
    _raise_for_status(int(status[0]))
    return alphas[0]


def opt_shortest_path_batch(problems: list, engine=None, **opt_kw) -> tuple:
    """Batch axis: problems = list of dicts {reftrack, normvectors, w_veh}.  Returns (alphas, status, infos)."""
    eng = engine or _engine.default_engine()
    packed = []
    for p in problems:
        ref = np.asarray(p["reftrack"], dtype=np.float64)
        nv = np.asarray(p["normvectors"], dtype=np.float64)
        _validate(ref, nv)
        packed.append(dict(reftrack=ref, normvec=nv, scaling=None, kappa_bound=1.0, w_veh=p["w_veh"]))
    alphas, _, status, infos = eng.solve_batch(packed, objective=_engine.OBJ_SHORTEST_PATH, **opt_kw)
    return alphas, status, infos


raise_for_status = _raise_for_status
