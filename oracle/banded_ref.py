"""
ORACLE -- TEST INFRASTRUCTURE ONLY (header of oracle/tph_ref.py applies).  PARITY UNPINNED by the reference.

ctypes binding of oracle/banded_qp.c: "CPU-B", the structure-exploiting scalar CPU solver of the reference's hot path
(tph.opt_min_curv -> quadprog, call sites [REF main_globaltraj.py:264-271, 344-350]), one problem per host thread.  Used as
bench.py's best-effort CPU baseline and, in tests, as a third route to alpha at sizes where the dense oracle takes a minute
per problem.  tests/test_oracle.py pins it against the dense-faithful oracle first.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_NATIVE = None
MIN_N = 4 * 72 + 4      # the bordered band of the C solver needs a ring much longer than its band (BE = 36)


def build():
    subprocess.run(["make", "-s", "-C", _HERE, "libbanded_qp.so"], check=True)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libbanded_qp.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "banded_qp.c")):
            build()
        _LIB = ctypes.CDLL(path)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        _LIB.bqp_solve_batch.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp, ip, ip,
                                         ctypes.c_int]
        _LIB.bqp_solve_batch.restype = ctypes.c_int
    return _LIB


def _lib_native():
    """The same source built on THIS host with -O3 -march=native (bench.py's CPU baseline; never shipped between machines)."""
    global _LIB_NATIVE
    if _LIB_NATIVE is None:
        path = os.path.join(_HERE, "libbanded_qp_native.so")
        subprocess.run(["make", "-s", "-B", "-C", _HERE, "libbanded_qp_native.so"], check=True)
        _LIB_NATIVE = ctypes.CDLL(path)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        _LIB_NATIVE.bqp_solve_batch.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp, ip, ip,
                                                ctypes.c_int]
        _LIB_NATIVE.bqp_solve_batch.restype = ctypes.c_int
    return _LIB_NATIVE


def solve_batch(reftrack, normvec, scaling, kappa_bound, w_veh, nthreads=0, native=False):
    """reftrack [B, n, 4], normvec [B, n, 2], scaling [B, n] or None.  Returns (alpha [B, n], curv_err [B], status [B],
    iters [B, 2] = (interior-point iterations, active-set rounds), threads used).  status: 0 ok, 1 infeasible widths, 2 not
    positive definite, 3 iteration cap, 4 bad input / ring too short, 6 a curvature row is violated at the box optimum."""
    ref = np.ascontiguousarray(reftrack, dtype=np.float64)
    nv = np.ascontiguousarray(normvec, dtype=np.float64)
    bsz, n = ref.shape[0], ref.shape[1]
    sc = None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float64)
    alpha = np.zeros((bsz, n))
    curv = np.zeros(bsz)
    status = np.zeros(bsz, dtype=np.int32)
    iters = np.zeros((bsz, 2), dtype=np.int32)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    used = (_lib_native() if native else _lib()).bqp_solve_batch(bsz, n, ref.ctypes.data_as(dp), nv.ctypes.data_as(dp),
                                  sc.ctypes.data_as(dp) if sc is not None else None, float(kappa_bound), float(w_veh),
                                  alpha.ctypes.data_as(dp), curv.ctypes.data_as(dp), status.ctypes.data_as(ip),
                                  iters.ctypes.data_as(ip), int(nthreads))
    return alpha, curv, status, iters, used
