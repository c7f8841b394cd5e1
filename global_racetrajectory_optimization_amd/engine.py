"""
ctypes host binding of libmcq.so (C ABI in include/mcq.h) -- the MI355X minimum-curvature QP engine.

This is plumbing, not a fallback layer: the only implementation behind these calls is the hand-written HIP library
built by csrc/build.sh for gfx950.  If it is missing or cannot be loaded the import of the engine fails loudly;
there is no CPU path in the product (the CPU oracle lives in /oracle and is test infrastructure only).

Mirrors the reference-side interface for the hot path:
    solve_batch(problems)          -> what tph.opt_min_curv.opt_min_curv binds  [REF main_globaltraj.py:264-271]
    Engine.solve_device(...)       -> device-resident batch entry (bench, IQP driver, multi-GPU shards)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libmcq.so")

STATUS_OK, STATUS_INFEASIBLE, STATUS_NOT_PD, STATUS_ITER_CAP, STATUS_BAD_INPUT, STATUS_KAPPA_INFEASIBLE, \
    STATUS_KAPPA_ACTIVE, STATUS_RING_OVERFLOW, STATUS_KAPPA_NO_SLOT = range(9)

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


class McqProblem(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int), ("reftrack", _dp), ("normvec", _dp), ("scaling", _dp),
                ("kappa_bound", ctypes.c_double), ("w_veh", ctypes.c_double)]


# the same record as a numpy structured type (field offsets taken from the ctypes layout): a batch's records are filled column by
# column -- one ctypes attribute store per field and track, as the per-track loop did it, costs 12 ms for 1024 tracks
_PROBLEM_DTYPE = np.dtype({"names": ["n", "reftrack", "normvec", "scaling", "kappa_bound", "w_veh"],
                           "formats": ["<i4", "<u8", "<u8", "<u8", "<f8", "<f8"],
                           "offsets": [getattr(McqProblem, f).offset for f in ("n", "reftrack", "normvec", "scaling", "kappa_bound", "w_veh")],
                           "itemsize": ctypes.sizeof(McqProblem)})


def _addr(a):
    return 0 if a is None else a.__array_interface__["data"][0]


def _addrs(arrays):
    """Addresses of a batch's per-track arrays: a list of arrays (None = NULL), or ONE stacked C-contiguous array [B, ...] (row k is
    track k: base + k * stride, no per-track Python work), or None (all NULL)."""
    if arrays is None:
        return 0
    if isinstance(arrays, np.ndarray):
        return arrays.__array_interface__["data"][0] + np.arange(arrays.shape[0], dtype=np.uint64) * np.uint64(arrays.strides[0])
    return [_addr(a) for a in arrays]


def problem_records(refs, nvs, scs, kappa_bound, w_veh):
    """mcq_problem records of a batch: refs / nvs / scs are lists of C-contiguous float64 arrays (entries of nvs / scs may be None =
    NULL) or stacked arrays [B, n, 4] / [B, n, 2] / [B, n] (see _addrs), kappa_bound / w_veh scalars or per-track sequences.  Returns
    (records, pointer for the C ABI); the caller keeps the arrays alive."""
    rec = np.zeros(len(refs), dtype=_PROBLEM_DTYPE)
    rec["n"] = refs.shape[1] if isinstance(refs, np.ndarray) else [r.shape[0] for r in refs]
    rec["reftrack"] = _addrs(refs)
    rec["normvec"] = _addrs(nvs)
    rec["scaling"] = _addrs(scs)
    rec["kappa_bound"] = kappa_bound
    rec["w_veh"] = w_veh
    return rec, ctypes.cast(rec.ctypes.data, ctypes.POINTER(McqProblem))


OBJ_MIN_CURV, OBJ_SHORTEST_PATH = 0, 1      # mcq_opts.objective (include/mcq.h)
ALG_DEFAULT, ALG_GI = 0, 1                  # mcq_opts.algorithm


class McqOpts(ctypes.Structure):
    _fields_ = [("algorithm", ctypes.c_int), ("max_ipm_iter", ctypes.c_int), ("max_as_iter", ctypes.c_int),
                ("refine_steps", ctypes.c_int), ("check_kappa", ctypes.c_int), ("objective", ctypes.c_int),
                ("warm_start", ctypes.c_int)]


class McqVelOpts(ctypes.Structure):
    _fields_ = [("dyn_model_exp", ctypes.c_double), ("filt_window", ctypes.c_int), ("reserved_", ctypes.c_int), ("mu", ctypes.c_void_p)]


class McqInfo(ctypes.Structure):
    _fields_ = [("ipm_iters", ctypes.c_int), ("as_iters", ctypes.c_int), ("n_active_box", ctypes.c_int),
                ("n_active_kappa", ctypes.c_int), ("kappa_max", ctypes.c_double), ("kkt_res", ctypes.c_double),
                ("ticks", ctypes.c_longlong * 8), ("refine_rounds", ctypes.c_int), ("second_attempt", ctypes.c_int),
                ("f32_factorisations", ctypes.c_int), ("gi_iters", ctypes.c_int)]


class McqIqpStats(ctypes.Structure):
    _fields_ = [("rounds", ctypes.c_int), ("qp_solves", ctypes.c_int), ("solver_ms", ctypes.c_float * 16),
                ("fallbacks", ctypes.c_int * 16), ("timed", ctypes.c_int)]


IQP_TRACE = 16      # MCQ_IQP_TRACE

EXPORTED_SYMBOLS = ("mcq_create", "mcq_destroy", "mcq_last_error", "mcq_default_opts", "mcq_solve_batch", "mcq_solve_host",
                    "mcq_iqp_device", "mcq_iqp_batch", "mcq_iqp_set_round_callback", "mcq_host_alloc", "mcq_host_free",
                    "mcq_solve_device", "mcq_solve_device_f32", "mcq_solve_device_f32_rows", "mcq_solve_batch_f32",
                    "mcq_solve_host_pipelined", "mcq_solve_device_stream", "mcq_solve_device_ragged", "mcq_solve_device_ragged_params", "mcq_prep_device", "mcq_relinearise_device",
                    "mcq_vel_profile_device", "mcq_vel_profile_device_ragged", "mcq_vel_profile_device_opts", "mcq_raceline_device", "mcq_normals_crossing_device",
                    "mcq_device_alloc",
                    "mcq_device_free", "mcq_copy_to_device", "mcq_copy_to_host", "mcq_sync", "mcq_stream",
                    "mcq_last_timing", "mcq_timing_begin", "mcq_timing_end", "mcq_workspace_bytes",
                    "mcq_comm_unique_id", "mcq_comm_init", "mcq_comm_allgather", "mcq_comm_wait", "mcq_comm_world", "mcq_comm_destroy",
                    "mcq_les_scalings", "mcq_last_upload_was_direct")


IQP_ROUND_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int))


class EngineError(RuntimeError):
    pass


def load_library(path=None):
    """Load libmcq.so and declare the C ABI.  Raises EngineError if the HIP library is absent -- never falls back."""
    path = path or os.environ.get("MCQ_LIB") or DEFAULT_LIB
    if not os.path.exists(path):
        raise EngineError("MI355X engine library not found at %s -- build it with "
                          "global_racetrajectory_optimization_amd/csrc/build.sh (hipcc, gfx950)" % path)
    lib = ctypes.CDLL(path)
    vp = ctypes.c_void_p
    lib.mcq_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.mcq_create.restype = ctypes.c_int
    lib.mcq_destroy.argtypes = [vp]
    lib.mcq_destroy.restype = None
    lib.mcq_last_error.argtypes = []
    lib.mcq_last_error.restype = ctypes.c_char_p
    lib.mcq_default_opts.argtypes = [ctypes.POINTER(McqOpts)]
    lib.mcq_default_opts.restype = None
    lib.mcq_solve_batch.argtypes = [vp, ctypes.POINTER(McqProblem), ctypes.c_int, ctypes.POINTER(McqOpts), _dp, _dp,
                                    _ip, ctypes.POINTER(McqInfo)]
    lib.mcq_solve_batch.restype = ctypes.c_int
    lib.mcq_solve_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_double, ctypes.c_double,
                                     ctypes.POINTER(McqOpts), vp, vp, vp, vp]
    lib.mcq_solve_device.restype = ctypes.c_int
    lib.mcq_solve_host.argtypes = lib.mcq_solve_device.argtypes
    lib.mcq_solve_host.restype = ctypes.c_int
    lib.mcq_iqp_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(McqOpts), vp, vp, vp,
                                   vp, vp, vp, ctypes.POINTER(McqIqpStats)]
    lib.mcq_iqp_device.restype = ctypes.c_int
    lib.mcq_iqp_batch.argtypes = [vp, ctypes.POINTER(McqProblem), ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                                  ctypes.c_int, ctypes.POINTER(McqOpts), ctypes.c_int, _dp, _dp, _dp, _ip, _dp, _ip, _ip, _dp,
                                  ctypes.POINTER(McqIqpStats)]
    lib.mcq_iqp_batch.restype = ctypes.c_int
    lib.mcq_iqp_set_round_callback.argtypes = [vp, IQP_ROUND_CB, vp]
    lib.mcq_iqp_set_round_callback.restype = ctypes.c_int
    lib.mcq_last_upload_was_direct.argtypes = [vp]
    lib.mcq_last_upload_was_direct.restype = ctypes.c_int
    lib.mcq_les_scalings.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int]
    lib.mcq_les_scalings.restype = ctypes.c_int
    lib.mcq_host_alloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.mcq_host_alloc.restype = ctypes.c_int
    lib.mcq_host_free.argtypes = [vp, vp]
    lib.mcq_host_free.restype = ctypes.c_int
    lib.mcq_solve_device_f32.argtypes = lib.mcq_solve_device.argtypes
    lib.mcq_solve_device_f32.restype = ctypes.c_int
    lib.mcq_solve_device_f32_rows.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_double, ctypes.c_double,
                                              ctypes.POINTER(McqOpts), vp, vp, vp, vp]
    lib.mcq_solve_device_f32_rows.restype = ctypes.c_int
    lib.mcq_solve_batch_f32.argtypes = lib.mcq_solve_device_f32_rows.argtypes
    lib.mcq_solve_batch_f32.restype = ctypes.c_int
    lib.mcq_solve_host_pipelined.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_double, ctypes.c_double,
                                             ctypes.POINTER(McqOpts), vp, vp, vp]
    lib.mcq_solve_host_pipelined.restype = ctypes.c_int
    lib.mcq_solve_device_stream.argtypes = lib.mcq_solve_host_pipelined.argtypes
    lib.mcq_solve_device_stream.restype = ctypes.c_int
    lib.mcq_solve_device_ragged.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_double,
                                            ctypes.c_double, ctypes.POINTER(McqOpts), vp, vp, vp, vp]
    lib.mcq_solve_device_ragged.restype = ctypes.c_int
    lib.mcq_solve_device_ragged_params.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_double,
                                                   ctypes.c_double, vp, vp, ctypes.POINTER(McqOpts), vp, vp, vp, vp]
    lib.mcq_solve_device_ragged_params.restype = ctypes.c_int
    lib.mcq_prep_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp]
    lib.mcq_prep_device.restype = ctypes.c_int
    lib.mcq_relinearise_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_double,
                                           ctypes.c_double, vp, vp, vp, vp]
    lib.mcq_relinearise_device.restype = ctypes.c_int
    lib.mcq_vel_profile_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, vp,
                                           ctypes.c_int, vp, vp, vp, ctypes.c_double, vp, vp]
    lib.mcq_vel_profile_device.restype = ctypes.c_int
    lib.mcq_vel_profile_device_ragged.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, vp,
                                                  ctypes.c_int, vp, vp, vp, ctypes.c_double, vp, vp]
    lib.mcq_vel_profile_device_ragged.restype = ctypes.c_int
    lib.mcq_vel_profile_device_opts.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, vp,
                                                ctypes.c_int, vp, vp, vp, ctypes.POINTER(McqVelOpts), vp, vp]
    lib.mcq_vel_profile_device_opts.restype = ctypes.c_int
    lib.mcq_raceline_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_double, ctypes.c_int, vp, vp,
                                        vp, vp, vp, vp]
    lib.mcq_raceline_device.restype = ctypes.c_int
    lib.mcq_normals_crossing_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]
    lib.mcq_normals_crossing_device.restype = ctypes.c_int
    lib.mcq_device_alloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.mcq_device_alloc.restype = ctypes.c_int
    lib.mcq_device_free.argtypes = [vp, vp]
    lib.mcq_device_free.restype = ctypes.c_int
    lib.mcq_copy_to_device.argtypes = [vp, vp, vp, ctypes.c_size_t]
    lib.mcq_copy_to_device.restype = ctypes.c_int
    lib.mcq_copy_to_host.argtypes = [vp, vp, vp, ctypes.c_size_t]
    lib.mcq_copy_to_host.restype = ctypes.c_int
    lib.mcq_sync.argtypes = [vp]
    lib.mcq_sync.restype = ctypes.c_int
    lib.mcq_stream.argtypes = [vp]
    lib.mcq_stream.restype = vp
    lib.mcq_last_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float * 5)]
    lib.mcq_last_timing.restype = ctypes.c_int
    lib.mcq_timing_begin.argtypes = [vp]
    lib.mcq_timing_begin.restype = ctypes.c_int
    lib.mcq_timing_end.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
    lib.mcq_timing_end.restype = ctypes.c_int
    lib.mcq_workspace_bytes.argtypes = [vp]
    lib.mcq_workspace_bytes.restype = ctypes.c_longlong
    lib.mcq_comm_unique_id.argtypes = [ctypes.c_char_p]
    lib.mcq_comm_unique_id.restype = ctypes.c_int
    lib.mcq_comm_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    lib.mcq_comm_init.restype = ctypes.c_int
    lib.mcq_comm_allgather.argtypes = [vp, vp, vp, ctypes.c_size_t, ctypes.c_int]
    lib.mcq_comm_allgather.restype = ctypes.c_int
    lib.mcq_comm_wait.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    lib.mcq_comm_wait.restype = ctypes.c_int
    lib.mcq_comm_world.argtypes = [vp, _ip, _ip]
    lib.mcq_comm_world.restype = ctypes.c_int
    lib.mcq_comm_destroy.argtypes = [vp]
    lib.mcq_comm_destroy.restype = ctypes.c_int
    return lib


def _as_dp(a):
    return a.ctypes.data_as(_dp)


F32_ABSOLUTE, F32_INCREMENTS = 0, 1      # MCQ_F32_* (include/mcq.h): layout of the coordinate columns of float rows


def rows_to_increments(reftrack):
    """fp64 rows [..., n, 4] = [x, y, w_r, w_l] -> (float32 rows [..., n, 4] = [x_{i+1} - x_i, y_{i+1} - y_i, w_r, w_l] around the
    ring, float64 origin [..., 2] = the first waypoint): the MCQ_F32_INCREMENTS layout of the fp32 entries.  The differences are
    taken in fp64 and rounded once."""
    ref = np.asarray(reftrack, dtype=np.float64)
    out = np.empty(ref.shape, dtype=np.float32)
    out[..., :2] = np.roll(ref[..., :2], -1, axis=-2) - ref[..., :2]
    out[..., 2:] = ref[..., 2:]
    return out, np.ascontiguousarray(ref[..., 0, :2])


def increments_to_rows(rows32, origin=None):
    """What the device rebuilds from MCQ_F32_INCREMENTS rows (mcq_widen_rows_kernel), in numpy: fp64 running sum from the origin with
    the closure defect of the float increments spread evenly over the ring.  For tests and for callers that want the reference
    line the engine saw."""
    r = np.asarray(rows32, dtype=np.float64)
    n = r.shape[-2]
    d = r[..., :2] - r[..., :2].mean(axis=-2, keepdims=True)
    xy = np.concatenate((np.zeros(r.shape[:-2] + (1, 2)), np.cumsum(d, axis=-2)[..., :-1, :]), axis=-2)
    if origin is not None:
        xy = xy + np.asarray(origin, dtype=np.float64)[..., None, :]
    return np.concatenate((xy, r[..., 2:]), axis=-1)


class Engine:
    """One handle == one GPU + one HIP stream + its device workspace.  One host thread at a time."""

    def __init__(self, device_id=0, lib_path=None):
        self.lib = load_library(lib_path)
        h = ctypes.c_void_p()
        rc = self.lib.mcq_create(int(device_id), ctypes.byref(h))
        if rc != 0:
            raise EngineError("mcq_create(%d) failed: %s" % (device_id, self.lib.mcq_last_error().decode()))
        self.h = h
        self.device_id = int(device_id)
        self.lib_path = lib_path

    def last_upload_was_direct(self):
        """True if the last solve_batch / iqp_batch went to the device without the packing pass (uniform batch, rows back to back in host_array memory)."""
        return bool(self.lib.mcq_last_upload_was_direct(self.h))

    def close(self):
        if getattr(self, "h", None):
            for p in getattr(self, "_pinned", []):
                self.lib.mcq_host_free(self.h, p)
            self._pinned = []
            self.lib.mcq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _opts(self, algorithm=ALG_DEFAULT, max_ipm_iter=0, max_as_iter=0, refine_steps=-1, check_kappa=1, objective=OBJ_MIN_CURV,
              warm_start=0):
        """mcq_opts.  algorithm=ALG_GI: every problem through the engine's Goldfarb-Idnani path (quadprog's algorithm; a reference mode).
        $MCQ_ALGORITHM=gi selects it for callers that cannot pass options -- the drop-in package under the untouched main_globaltraj.py."""
        if int(algorithm) == ALG_DEFAULT and os.environ.get("MCQ_ALGORITHM", "").lower() in ("gi", "goldfarb-idnani", "quadprog"):
            algorithm = ALG_GI
        return McqOpts(int(algorithm), int(max_ipm_iter), int(max_as_iter), int(refine_steps), int(check_kappa),
                       int(objective), int(warm_start))

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError("%s failed (%d): %s" % (what, rc, self.lib.mcq_last_error().decode()))

    # ------------------------------------------------------------------------------------------------------------------
    def solve_batch(self, problems, **opt_kw):
        """problems: list of dicts {reftrack [n,4], normvec [n,2], scaling [n] or None, kappa_bound, w_veh}.

        Returns (alphas: list of [n] arrays, curv_err [B], status [B] int32, info: list of dicts).
        """
        bsz = len(problems)
        keep = []
        total = 0
        for k, p in enumerate(problems):
            ref = np.ascontiguousarray(p["reftrack"], dtype=np.float64)
            nv = None if p.get("normvec") is None else np.ascontiguousarray(p["normvec"], dtype=np.float64)
            n = ref.shape[0]
            if ref.ndim != 2 or ref.shape[1] != 4 or (nv is not None and nv.shape != (n, 2)):
                raise ValueError("reftrack must be [n,4] and normvec [n,2]")
            sc = p.get("scaling")
            if sc is not None:
                sc = np.ascontiguousarray(sc, dtype=np.float64)
                if sc.shape != (n,):
                    raise ValueError("scaling must be [n]")
            keep.append((ref, nv, sc))
            total += n
        rec, arr = problem_records([q[0] for q in keep], [q[1] for q in keep], [q[2] for q in keep],
                                   [float(p["kappa_bound"]) for p in problems], [float(p["w_veh"]) for p in problems])
        alpha = np.zeros(total)
        curv = np.zeros(bsz)
        status = np.zeros(bsz, dtype=np.int32)
        info = (McqInfo * bsz)()
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_batch(self.h, arr, bsz, ctypes.byref(opts), _as_dp(alpha), _as_dp(curv),
                                      status.ctypes.data_as(_ip), info)
        self._check(rc, "mcq_solve_batch")
        out, off = [], 0
        for ref, _, _ in keep:
            out.append(alpha[off:off + ref.shape[0]].copy())
            off += ref.shape[0]
        infos = [dict(ipm_iters=i.ipm_iters, as_iters=i.as_iters, n_active_box=i.n_active_box,
                      n_active_kappa=i.n_active_kappa, kappa_max=i.kappa_max, kkt_res=i.kkt_res,
                      ticks=list(i.ticks), refine_rounds=i.refine_rounds, second_attempt=i.second_attempt,
                      f32_factorisations=i.f32_factorisations, gi_iters=i.gi_iters) for i in info]
        return out, curv, status, infos

    # ------------------------------------------------------------------------------------------------------------------
    def solve_device(self, batch, n, d_reftrack, d_normvec, d_scaling, kappa_bound, w_veh, d_alpha, d_curv, d_status,
                     d_info=None, **opt_kw):
        """Device-resident batch (raw device pointers as ints, e.g. torch.Tensor.data_ptr()).  Asynchronous."""
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_device(self.h, int(batch), int(n), d_reftrack, d_normvec, d_scaling or None,
                                       float(kappa_bound), float(w_veh), ctypes.byref(opts), d_alpha, d_curv, d_status,
                                       d_info or None)
        self._check(rc, "mcq_solve_device")

    def solve_device_f32(self, batch, n, d_reftrack, d_normvec, d_scaling, kappa_bound, w_veh, d_alpha, d_curv, d_status,
                         d_info=None, **opt_kw):
        """As solve_device with float32 tracks and float32 alpha in HBM (mcq_solve_device_f32: fp32 at the boundary, fp64
        arithmetic inside; d_curv stays float64).  d_normvec / d_scaling may be None."""
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_device_f32(self.h, int(batch), int(n), d_reftrack, d_normvec or None, d_scaling or None,
                                           float(kappa_bound), float(w_veh), ctypes.byref(opts), d_alpha, d_curv, d_status,
                                           d_info or None)
        self._check(rc, "mcq_solve_device_f32")

    def solve_host(self, reftrack, normvec, scaling, kappa_bound, w_veh, alpha_out=None, **opt_kw):
        """Uniform batch from / to host arrays in one call (mcq_solve_host): reftrack [B, n, 4], normvec [B, n, 2] or None,
        scaling [B, n] or None -- float64, C-contiguous, ideally views of pinned memory (host_array).  Returns (alpha [B, n],
        curv_err [B], status [B], info as a structured array of McqInfo)."""
        ref = np.ascontiguousarray(reftrack, dtype=np.float64)
        bsz, n = ref.shape[0], ref.shape[1]
        nv = None if normvec is None else np.ascontiguousarray(normvec, dtype=np.float64)
        sc = None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float64)
        if alpha_out is None:
            alpha = np.empty((bsz, n))
        else:
            alpha = alpha_out
            if not isinstance(alpha, np.ndarray) or alpha.dtype != np.float64 or alpha.shape != (bsz, n) or not alpha.flags["C_CONTIGUOUS"] \
                    or not alpha.flags["WRITEABLE"]:
                raise ValueError("alpha_out must be a writeable C-contiguous float64 array of shape %r" % ((bsz, n),))
        curv = np.empty(bsz)
        status = np.empty(bsz, dtype=np.int32)
        info = (McqInfo * bsz)()
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_host(self.h, bsz, n, ref.ctypes.data, nv.ctypes.data if nv is not None else None,
                                     sc.ctypes.data if sc is not None else None, float(kappa_bound), float(w_veh),
                                     ctypes.byref(opts), alpha.ctypes.data, curv.ctypes.data, status.ctypes.data,
                                     ctypes.addressof(info))
        self._check(rc, "mcq_solve_host")
        return alpha, curv, status, info

    def solve_device_stream(self, batch, n, d_reftracks, d_normvecs, d_scalings, kappa_bound, w_veh, d_alphas, d_curvs, d_statuses, **opt_kw):
        """A stream of resident uniform batches on the engine's two compute streams (mcq_solve_device_stream): lists of raw device pointers, one
        entry per step (d_normvecs / d_scalings: None, or lists whose entries may be None).  Asynchronous: sync() waits for every step."""
        steps = len(d_reftracks)

        def arr(seq):
            if seq is None:
                return None
            a = (ctypes.c_void_p * steps)()
            for k, p in enumerate(seq):
                a[k] = p
            return a
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_device_stream(self.h, steps, int(batch), int(n), arr(d_reftracks), arr(d_normvecs), arr(d_scalings),
                                              float(kappa_bound), float(w_veh), ctypes.byref(opts), arr(d_alphas), arr(d_curvs), arr(d_statuses))
        self._check(rc, "mcq_solve_device_stream")

    def host_array(self, shape, dtype=np.float64):
        """numpy array backed by pinned (page-locked) host memory of the engine (mcq_host_alloc).  LIFETIME: the memory belongs to
        the engine -- it is released by host_free(array) or by close() / garbage collection of the engine, after which the array
        (and every view of it) must not be touched.  Keep the engine alive as long as the arrays are in use."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        self._check(self.lib.mcq_host_alloc(self.h, max(nbytes, 1), ctypes.byref(p)), "mcq_host_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def host_free(self, arr):
        """Release a host_array before the engine goes away (the array must not be used afterwards)."""
        addr = int(arr.ctypes.data)
        pinned = getattr(self, "_pinned", [])
        if addr not in pinned:
            raise ValueError("host_free: not the start of an array from host_array of this engine")
        pinned.remove(addr)
        self._check(self.lib.mcq_host_free(self.h, addr), "mcq_host_free")

    def solve_host_pipelined(self, reftracks, normvecs, scalings, kappa_bound, w_veh, alpha_outs, **opt_kw):
        """A stream of uniform host batches with the PCIe hidden behind the kernels (mcq_solve_host_pipelined): reftracks / normvecs /
        scalings / alpha_outs are lists (one entry per step) of C-contiguous float64 arrays [B, n, 4] / [B, n, 2] or None / [B, n] or
        None / [B, n] -- ideally host_array (pinned) memory; entries may repeat.  Returns (curv_err [steps, B], status [steps, B])."""
        steps = len(reftracks)
        bsz, n = reftracks[0].shape[0], reftracks[0].shape[1]

        keep = []

        def ptrs(seq, shape, allow_none, output=False):
            arr = (ctypes.c_void_p * steps)()
            for k, a in enumerate(seq):
                if a is None:
                    if not allow_none:
                        raise ValueError("missing buffer for step %d" % k)
                    arr[k] = None
                    continue
                if output:
                    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.shape != shape or not a.flags["C_CONTIGUOUS"] \
                            or not a.flags["WRITEABLE"]:
                        raise ValueError("step %d: expected a writeable C-contiguous float64 array of shape %r" % (k, shape))
                else:
                    a = np.ascontiguousarray(a, dtype=np.float64)       # a no-op for the pinned arrays this entry is meant for
                    if a.shape != shape:
                        raise ValueError("step %d: expected shape %r" % (k, shape))
                    keep.append(a)
                arr[k] = a.ctypes.data
            return arr
        p_ref = ptrs(reftracks, (bsz, n, 4), False)
        p_nv = ptrs(normvecs, (bsz, n, 2), True) if normvecs is not None else None
        p_sc = ptrs(scalings, (bsz, n), True) if scalings is not None else None
        p_al = ptrs(alpha_outs, (bsz, n), False, output=True)
        curv = np.zeros((steps, bsz))
        status = np.zeros((steps, bsz), dtype=np.int32)
        p_cu = (ctypes.c_void_p * steps)(*[curv[k].ctypes.data for k in range(steps)])
        p_st = (ctypes.c_void_p * steps)(*[status[k].ctypes.data for k in range(steps)])
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_host_pipelined(self.h, steps, bsz, n, p_ref, p_nv, p_sc, float(kappa_bound), float(w_veh),
                                               ctypes.byref(opts), p_al, p_cu, p_st)
        self._check(rc, "mcq_solve_host_pipelined")
        return curv, status

    def solve_device_f32_rows(self, batch, n, layout, d_rows, d_origin, kappa_bound, w_veh, d_alpha, d_curv, d_status, d_info=None,
                              **opt_kw):
        """Device-resident float rows in either layout (F32_ABSOLUTE / F32_INCREMENTS; mcq_solve_device_f32_rows): d_rows
        [batch, n, 4] float32, d_origin [batch, 2] float64 or None, d_alpha [batch, n] float32.  Asynchronous."""
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_device_f32_rows(self.h, int(batch), int(n), int(layout), d_rows, d_origin or None, float(kappa_bound),
                                                float(w_veh), ctypes.byref(opts), d_alpha, d_curv, d_status, d_info or None)
        self._check(rc, "mcq_solve_device_f32_rows")

    def solve_batch_f32(self, rows32, origin, kappa_bound, w_veh, layout=F32_INCREMENTS, **opt_kw):
        """Host-buffer fp32 entry (mcq_solve_batch_f32): rows32 [B, n, 4] float32 in `layout`, origin [B, 2] float64 or None.
        Returns (alpha float32 [B, n], curv_err [B], status [B], info as a ctypes array of McqInfo)."""
        rows = np.ascontiguousarray(rows32, dtype=np.float32)
        bsz, n = rows.shape[0], rows.shape[1]
        org = None if origin is None else np.ascontiguousarray(origin, dtype=np.float64)
        if rows.ndim != 3 or rows.shape[2] != 4 or (org is not None and org.shape != (bsz, 2)):
            raise ValueError("rows32 must be [B, n, 4] and origin [B, 2]")
        alpha = np.zeros((bsz, n), dtype=np.float32)
        curv = np.zeros(bsz)
        status = np.zeros(bsz, dtype=np.int32)
        info = (McqInfo * bsz)()
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_batch_f32(self.h, bsz, n, int(layout), rows.ctypes.data, org.ctypes.data if org is not None else None,
                                          float(kappa_bound), float(w_veh), ctypes.byref(opts), alpha.ctypes.data, curv.ctypes.data,
                                          status.ctypes.data, ctypes.addressof(info))
        self._check(rc, "mcq_solve_batch_f32")
        return alpha, curv, status, info

    def iqp_batch(self, tracks, kappa_bound, w_veh, stepsize_interp, iters_min=3, curv_error_allowed=0.01, max_rounds=50,
                  nmax=None, timed=False, out=None, **opt_kw):
        """tph.iqp_handler for a batch of tracks as ONE engine call (mcq_iqp_batch): tracks = list of dicts {reftrack [N,4],
        normvectors [N,2], scaling [N] or None}.  Returns a dict: alpha / reftrack / normvectors (lists of the final arrays),
        n, curv_err, status, rounds (arrays [B]), curv_trace [B, IQP_TRACE], stats (rounds, qp_solves, and with timed=True
        solver_ms / fallbacks per round).
        out (optional): dict with preallocated 'alpha' [B, nmax], 'reftrack' [B, nmax, 4], 'normvectors' [B, nmax, 2] float64 arrays
        (e.g. from host_array: page-locked) that receive the end states -- a caller that runs batch after batch keeps them; the
        returned per-track arrays are views into them.  Without it fresh arrays are allocated AND touched here: a device-to-host
        copy into never-touched pageable memory runs at a tenth of the PCIe rate (page faults inside the copy)."""
        stacked = isinstance(tracks, dict)
        if stacked:
            # a uniform batch as ONE dict of stacked arrays {reftrack [B,N,4], normvectors [B,N,2], scaling [B,N] or None}: no per-track
            # Python work at all (building 1024 records track by track costs 4-12 ms of a 40 ms call)
            refs = np.ascontiguousarray(tracks["reftrack"], dtype=np.float64)
            nvs = np.ascontiguousarray(tracks["normvectors"], dtype=np.float64)
            scs = None if tracks.get("scaling") is None else np.ascontiguousarray(tracks["scaling"], dtype=np.float64)
            if refs.ndim != 3 or refs.shape[2] != 4 or nvs.shape != refs.shape[:2] + (2,) or (scs is not None and scs.shape != refs.shape[:2]):
                raise ValueError("stacked tracks: reftrack must be [B,n,4], normvectors [B,n,2], scaling [B,n]")
            bsz = refs.shape[0]
        else:
            bsz = len(tracks)
            refs = [np.ascontiguousarray(t["reftrack"], dtype=np.float64) for t in tracks]
            nvs = [np.ascontiguousarray(t["normvectors"], dtype=np.float64) for t in tracks]
            scs = [None if t.get("scaling") is None else np.ascontiguousarray(t["scaling"], dtype=np.float64) for t in tracks]
        if nmax is None:
            # capacity for the re-sampled rings: the raceline is never much longer than the polygon through the reference
            # points; 30 % + 16 points of headroom (a ring that outgrows it is reported per track, not truncated)
            if stacked or len({r.shape for r in refs}) == 1:           # uniform batch: one vectorised pass over all tracks
                xy = refs[:, :, :2] if stacked else np.stack([r[:, :2] for r in refs])
                seg = xy - np.roll(xy, -1, axis=1)
                lengths = np.sqrt(np.einsum("bnk,bnk->bn", seg, seg)).sum(axis=1)
            else:
                lengths = np.array([np.hypot(np.diff(r[:, 0], append=r[0, 0]), np.diff(r[:, 1], append=r[0, 1])).sum() for r in refs])
            finite = lengths[np.isfinite(lengths)]          # (a track with non-finite rows is reported by the engine: status 4)
            longest = float(finite.max()) if finite.size else 0.0
            nmax = max(refs.shape[1] if stacked else max(r.shape[0] for r in refs), int(np.ceil(1.3 * longest / stepsize_interp)) + 16)
        for k in range(0 if stacked else bsz):
            n = refs[k].shape[0]
            if refs[k].ndim != 2 or refs[k].shape[1] != 4 or nvs[k].shape != (n, 2) or (scs[k] is not None and scs[k].shape != (n,)):
                raise ValueError("reftrack must be [n,4], normvectors [n,2], scaling [n]")
        rec, arr = problem_records(refs, nvs, scs, float(kappa_bound), float(w_veh))
        def _out(key, shape):
            if out is not None and key in out:
                a = out[key]
                if a.dtype != np.float64 or not a.flags["C_CONTIGUOUS"] or a.shape != shape:
                    raise ValueError("out[%r] must be a C-contiguous float64 array of shape %r" % (key, shape))
                return a
            a = np.empty(shape)
            a.fill(0.0)              # touch the pages before the driver copies into them
            return a
        alpha = _out("alpha", (bsz, nmax))
        ref_o = _out("reftrack", (bsz, nmax, 4))
        nv_o = _out("normvectors", (bsz, nmax, 2))
        n_o = np.zeros(bsz, dtype=np.int32)
        curv = np.zeros(bsz)
        status = np.zeros(bsz, dtype=np.int32)
        rounds = np.zeros(bsz, dtype=np.int32)
        trace = np.zeros((bsz, IQP_TRACE))
        st = McqIqpStats()
        st.timed = 1 if timed else 0
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_iqp_batch(self.h, arr, bsz, float(stepsize_interp), int(iters_min), float(curv_error_allowed),
                                    int(max_rounds), ctypes.byref(opts), int(nmax), _as_dp(alpha), _as_dp(ref_o), _as_dp(nv_o),
                                    n_o.ctypes.data_as(_ip), _as_dp(curv), status.ctypes.data_as(_ip),
                                    rounds.ctypes.data_as(_ip), _as_dp(trace), ctypes.byref(st))
        self._check(rc, "mcq_iqp_batch")
        stats = dict(rounds=int(st.rounds), qp_solves=int(st.qp_solves), nmax=int(nmax))
        if timed:
            stats["solver_ms"] = [float(st.solver_ms[k]) for k in range(min(st.rounds, 16))]
            stats["fallbacks"] = [int(st.fallbacks[k]) for k in range(min(st.rounds, 16))]
        return dict(alpha=[alpha[k, :n_o[k]] for k in range(bsz)], reftrack=[ref_o[k, :n_o[k]] for k in range(bsz)],
                    normvectors=[nv_o[k, :n_o[k]] for k in range(bsz)], n=n_o, curv_err=curv, status=status, rounds=rounds,
                    curv_trace=trace, stats=stats)

    def solve_uniform_f32(self, reftrack, normvec, scaling, kappa_bound, w_veh, **opt_kw):
        """Host convenience around solve_device_f32 for a uniform-n batch: reftrack [B,n,4] (cast to float32), normvec
        [B,n,2] or None (derived on the device), scaling [B,n] or None.  Returns (alpha float32 [B,n], curv_err [B],
        status [B], info [B])."""
        ref = np.ascontiguousarray(reftrack, dtype=np.float32)
        bsz, n = ref.shape[0], ref.shape[1]
        nv = None if normvec is None else np.ascontiguousarray(normvec, dtype=np.float32)
        sc = None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float32)
        bufs = []

        def up(a):
            p = self.alloc(a.nbytes)
            bufs.append(p)
            self.upload(p, a)
            return p

        try:
            d_ref = up(ref)
            d_nv = up(nv) if nv is not None else None
            d_sc = up(sc) if sc is not None else None
            d_alpha = self.alloc(bsz * n * 4); bufs.append(d_alpha)
            d_curv = self.alloc(bsz * 8); bufs.append(d_curv)
            d_status = self.alloc(bsz * 4); bufs.append(d_status)
            d_info = self.alloc(bsz * ctypes.sizeof(McqInfo)); bufs.append(d_info)
            self.solve_device_f32(bsz, n, d_ref, d_nv, d_sc, kappa_bound, w_veh, d_alpha, d_curv, d_status, d_info, **opt_kw)
            self.sync()
            alpha = self.download(d_alpha, (bsz, n), np.float32)
            curv = self.download(d_curv, (bsz,), np.float64)
            status = self.download(d_status, (bsz,), np.int32)
            raw = self.download(d_info, (bsz * ctypes.sizeof(McqInfo),), np.uint8)
        finally:
            for p in bufs:
                self.free(p)
        infos = (McqInfo * bsz).from_buffer_copy(raw.tobytes())
        info = [dict(ipm_iters=i.ipm_iters, as_iters=i.as_iters, n_active_box=i.n_active_box,
                     n_active_kappa=i.n_active_kappa, kappa_max=i.kappa_max, kkt_res=i.kkt_res) for i in infos]
        return alpha, curv, status, info

    def solve_device_ragged(self, batch, nmax, d_n, d_reftrack, d_normvec, d_scaling, kappa_bound, w_veh, d_alpha,
                            d_curv, d_status, d_info=None, **opt_kw):
        """Device-resident batch with per-problem waypoint counts d_n [batch] (int32, device); arrays strided by nmax."""
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_device_ragged(self.h, int(batch), int(nmax), d_n, d_reftrack, d_normvec,
                                              d_scaling or None, float(kappa_bound), float(w_veh), ctypes.byref(opts),
                                              d_alpha, d_curv, d_status, d_info or None)
        self._check(rc, "mcq_solve_device_ragged")

    def solve_device_ragged_params(self, batch, nmax, d_n, d_reftrack, d_normvec, d_scaling, kappa_bound, w_veh, d_kappa_list,
                                   d_w_veh_list, d_alpha, d_curv, d_status, d_info=None, **opt_kw):
        """solve_device_ragged with per-problem kappa_bound / w_veh (device arrays [batch] or None)."""
        opts = self._opts(**opt_kw)
        rc = self.lib.mcq_solve_device_ragged_params(self.h, int(batch), int(nmax), d_n, d_reftrack, d_normvec or None,
                                                     d_scaling or None, float(kappa_bound), float(w_veh), d_kappa_list or None,
                                                     d_w_veh_list or None, ctypes.byref(opts), d_alpha, d_curv, d_status,
                                                     d_info or None)
        self._check(rc, "mcq_solve_device_ragged_params")

    def prep_batch(self, reftracks):
        """Unit normals and spline scalings of the closed distance-scaled splines through a list of reference lines
        ([n,>=2] arrays), computed on the device (mcq_prep_device).  Returns (list of [n,2], list of [n])."""
        bsz = len(reftracks)
        ns = np.array([r.shape[0] for r in reftracks], dtype=np.int32)
        nmax = int(ns.max())
        ref = np.zeros((bsz, nmax, 4))
        for k, r in enumerate(reftracks):
            ref[k, :ns[k], :min(4, r.shape[1])] = np.asarray(r, dtype=np.float64)[:, :4]
        d_ref, d_n = self.alloc(ref.nbytes), self.alloc(ns.nbytes)
        d_nv, d_sc, d_st = self.alloc(bsz * nmax * 16), self.alloc(bsz * nmax * 8), self.alloc(bsz * 4)
        try:
            self.upload(d_ref, ref)
            self.upload(d_n, ns)
            self._check(self.lib.mcq_prep_device(self.h, bsz, nmax, d_n, d_ref, d_nv, d_sc, d_st), "mcq_prep_device")
            st = self.download(d_st, (bsz,), np.int32)
            nv = self.download(d_nv, (bsz, nmax, 2), np.float64)
            sc = self.download(d_sc, (bsz, nmax), np.float64)
        finally:
            for p in (d_ref, d_n, d_nv, d_sc, d_st):
                self.free(p)
        if np.any(st != 0):
            raise EngineError("mcq_prep_device: bad input in problem(s) %s" % np.nonzero(st)[0].tolist())
        return [nv[k, :ns[k]].copy() for k in range(bsz)], [sc[k, :ns[k]].copy() for k in range(bsz)]

    def vel_profile_batch(self, kappa, el_lengths, ggv, ax_max_machines, drag_coeff, m_veh, v_max, dyn_model_exp=1.0,
                          track_of=None, n_of_track=None, mu=None, filt_window=None):
        """ggv velocity profiles and lap times of a batch of variants on the device (mcq_vel_profile_device; with mu -- friction
        coefficient per waypoint, [tracks, n] -- or filt_window -- tph.conv_filt's odd moving-average width over the finished profile --
        mcq_vel_profile_device_opts).

        kappa, el_lengths: [tracks, n]; ggv: [batch, g, 3]; ax_max_machines: [batch, m, 2]; drag_coeff, m_veh, v_max: [batch];
        track_of: [batch] ints (row of kappa / el per variant) or None when tracks == batch; n_of_track: [tracks] valid entries
        per row (ragged tracks, mcq_vel_profile_device_ragged) or None (all rows full).  Returns (vx [batch, n], lap_time
        [batch])."""
        kappa = np.ascontiguousarray(kappa, dtype=np.float64)
        el = np.ascontiguousarray(el_lengths, dtype=np.float64)
        ggv = np.ascontiguousarray(ggv, dtype=np.float64)
        axm = np.ascontiguousarray(ax_max_machines, dtype=np.float64)
        bsz, n = ggv.shape[0], kappa.shape[1]
        scal = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (bsz,))) for a in (drag_coeff, m_veh, v_max)]
        # what tph.calc_vel_profile raises on (the kernel itself flags such a variant with lap_time = NaN)
        if np.any(ggv[:, -1, 0] < scal[2]):
            raise RuntimeError("ggv has to cover the entire velocity range of the car (i.e. >= v_max)!")
        if np.any(axm[:, -1, 0] < scal[2]):
            raise RuntimeError("ax_max_machines has to cover the entire velocity range of the car (i.e. >= v_max)!")
        if filt_window is not None and int(filt_window) % 2 == 0:
            raise RuntimeError("Window width of moving average filter must be odd!")
        if filt_window is not None and int(filt_window) > 1:
            # a window wider than the ring: the kernel flags the variant with lap_time = NaN (tph.conv_filt wraps the ring more than once
            # and returns something of another length) -- refused here instead of handed back as NaNs
            n_min = int(np.min(n_of_track)) if n_of_track is not None else n
            if int(filt_window) > n_min:
                raise ValueError("vel_profile_batch: filt_window %d is wider than the shortest profile (%d points)" % (int(filt_window), n_min))
        if mu is not None:
            mu = np.ascontiguousarray(mu, dtype=np.float64)
            if mu.shape != kappa.shape:
                raise RuntimeError("kappa and mu must have the same length!")
        tr = None if track_of is None else np.ascontiguousarray(track_of, dtype=np.int32)
        nt = None if n_of_track is None else np.ascontiguousarray(n_of_track, dtype=np.int32)
        ptrs = []

        def up(a):
            p = self.alloc(a.nbytes)
            ptrs.append(p)
            self.upload(p, a)
            return p
        try:
            d_k, d_e, d_g, d_a = up(kappa), up(el), up(ggv), up(axm)
            d_s = [up(a) for a in scal]
            d_t = up(tr) if tr is not None else None
            d_vx = self.alloc(bsz * n * 8); ptrs.append(d_vx)
            d_lt = self.alloc(bsz * 8); ptrs.append(d_lt)
            if mu is not None or filt_window is not None:
                vo = McqVelOpts(float(dyn_model_exp), int(filt_window or 0), 0, up(mu) if mu is not None else None)
                rc = self.lib.mcq_vel_profile_device_opts(self.h, bsz, n, n, up(nt) if nt is not None else None, d_t, d_k, d_e, d_g,
                                                          ggv.shape[1], d_a, axm.shape[1], d_s[0], d_s[1], d_s[2], ctypes.byref(vo),
                                                          d_vx, d_lt)
            elif nt is not None:
                rc = self.lib.mcq_vel_profile_device_ragged(self.h, bsz, n, up(nt), d_t, d_k, d_e, d_g, ggv.shape[1], d_a,
                                                            axm.shape[1], d_s[0], d_s[1], d_s[2], float(dyn_model_exp),
                                                            d_vx, d_lt)
            else:
                rc = self.lib.mcq_vel_profile_device(self.h, bsz, n, n, d_t, d_k, d_e, d_g, ggv.shape[1], d_a, axm.shape[1],
                                                     d_s[0], d_s[1], d_s[2], float(dyn_model_exp), d_vx, d_lt)
            self._check(rc, "mcq_vel_profile_device")
            return self.download(d_vx, (bsz, n), np.float64), self.download(d_lt, (bsz,), np.float64)
        finally:
            for p in ptrs:
                self.free(p)

    def normals_crossing_batch(self, reftracks, normvecs, horizon=10):
        """tph.check_normals_crossing of a list of tracks on the device [REF helper_funcs_glob/src/prep_track.py:57-59].
        Returns an int32 array: 1 crossing, 0 none, -1 where tph would raise (horizon >= n)."""
        bsz = len(reftracks)
        ns = np.array([np.asarray(r).shape[0] for r in reftracks], dtype=np.int32)
        nmax = int(ns.max())
        ref = np.zeros((bsz, nmax, 4))
        nv = np.zeros((bsz, nmax, 2))
        for k in range(bsz):
            ref[k, :ns[k]] = np.asarray(reftracks[k], dtype=np.float64)[:, :4]
            nv[k, :ns[k]] = normvecs[k]
        ptrs = []
        try:
            for a in (ref, nv, ns):
                ptrs.append(self.alloc(a.nbytes))
                self.upload(ptrs[-1], a)
            ptrs.append(self.alloc(bsz * 4))
            rc = self.lib.mcq_normals_crossing_device(self.h, bsz, nmax, ptrs[2], ptrs[0], ptrs[1], int(horizon), ptrs[3])
            self._check(rc, "mcq_normals_crossing_device")
            return self.download(ptrs[3], (bsz,), np.int32)
        finally:
            for p in ptrs:
                self.free(p)

    def raceline_batch(self, reftracks, normvecs, alphas, stepsize, mmax=None):
        """tph.create_raceline + tph.calc_head_curv_an of a list of tracks on the device (mcq_raceline_device)
        [REF main_globaltraj.py:371-387].  reftracks [n_k, >=2], normvecs [n_k, 2], alphas [n_k].  Returns a dict of padded
        arrays: xy [B, mmax, 2], psi / kappa / el_lengths [B, mmax], m [B] (valid entries per row), status [B]."""
        bsz = len(reftracks)
        ns = np.array([np.asarray(r).shape[0] for r in reftracks], dtype=np.int32)
        nmax = int(ns.max())
        ref = np.zeros((bsz, nmax, 4))
        nv = np.zeros((bsz, nmax, 2))
        al = np.zeros((bsz, nmax))
        for k in range(bsz):
            r = np.asarray(reftracks[k], dtype=np.float64)
            ref[k, :ns[k], :min(4, r.shape[1])] = r[:, :4]
            nv[k, :ns[k]] = normvecs[k]
            al[k, :ns[k]] = alphas[k]
        if mmax is None:      # polygon length + the widest shift bounds the raceline length from above generously
            per = [float(np.sum(np.hypot(*np.diff(np.vstack((r[:, :2], r[:1, :2])), axis=0).T)) + 8.0 * np.sum(np.abs(a)))
                   for r, a in zip(reftracks, alphas)]
            mmax = int(max(per) / float(stepsize) * 1.25) + 16
        ptrs = []

        def up(a):
            p = self.alloc(a.nbytes)
            ptrs.append(p)
            self.upload(p, a)
            return p

        def new(nbytes):
            p = self.alloc(nbytes)
            ptrs.append(p)
            return p
        try:
            d_ref, d_nv, d_al, d_n = up(ref), up(nv), up(al), up(ns)
            d_xy, d_psi, d_k, d_el = new(bsz * mmax * 16), new(bsz * mmax * 8), new(bsz * mmax * 8), new(bsz * mmax * 8)
            d_m, d_st = new(bsz * 4), new(bsz * 4)
            rc = self.lib.mcq_raceline_device(self.h, bsz, nmax, d_n, d_ref, d_nv, d_al, float(stepsize), int(mmax), d_xy,
                                              d_psi, d_k, d_el, d_m, d_st)
            self._check(rc, "mcq_raceline_device")
            return dict(xy=self.download(d_xy, (bsz, mmax, 2), np.float64), psi=self.download(d_psi, (bsz, mmax), np.float64),
                        kappa=self.download(d_k, (bsz, mmax), np.float64),
                        el_lengths=self.download(d_el, (bsz, mmax), np.float64),
                        m=self.download(d_m, (bsz,), np.int32), status=self.download(d_st, (bsz,), np.int32))
        finally:
            for p in ptrs:
                self.free(p)

    def relinearise_device(self, batch, nmax, d_n_in, d_ref_in, d_nv_in, d_alpha, d_live, alpha_scale, stepsize,
                           d_ref_out, d_nv_out, d_n_out, d_status):
        """IQP glue between two passes, on the device (mcq_relinearise_device in include/mcq.h).  Asynchronous."""
        rc = self.lib.mcq_relinearise_device(self.h, int(batch), int(nmax), d_n_in, d_ref_in, d_nv_in, d_alpha,
                                             d_live or None, float(alpha_scale), float(stepsize), d_ref_out, d_nv_out,
                                             d_n_out, d_status)
        self._check(rc, "mcq_relinearise_device")

    def set_iqp_round_callback(self, fn):
        """fn(round, curv_err [batch] ndarray, live [batch] ndarray) after every QP pass of mcq_iqp_batch / mcq_iqp_device, or None to remove
        it (the per-iteration print_debug lines of tph.iqp_handler, as they happen)."""
        if fn is None:
            self._iqp_cb = None
            self._check(self.lib.mcq_iqp_set_round_callback(self.h, ctypes.cast(None, IQP_ROUND_CB), None), "mcq_iqp_set_round_callback")
            return

        def tramp(_user, rnd, batch, curv, live):
            fn(int(rnd), np.ctypeslib.as_array(curv, shape=(batch,)).copy(), np.ctypeslib.as_array(live, shape=(batch,)).copy())
        self._iqp_cb = IQP_ROUND_CB(tramp)          # kept alive while registered
        self._check(self.lib.mcq_iqp_set_round_callback(self.h, self._iqp_cb, None), "mcq_iqp_set_round_callback")

    # ---- device memory plumbing (mcq_device_alloc & co): numpy in, numpy out, raw device pointers as ints ----------------
    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self.lib.mcq_device_alloc(self.h, int(nbytes), ctypes.byref(p)), "mcq_device_alloc")
        return p.value

    def free(self, ptr):
        self._check(self.lib.mcq_device_free(self.h, ptr), "mcq_device_free")

    def upload(self, ptr, arr, offset_bytes=0):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.mcq_copy_to_device(self.h, int(ptr) + int(offset_bytes), arr.ctypes.data, arr.nbytes), "mcq_copy_to_device")

    def download(self, ptr, shape, dtype, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes >= (1 << 20):
            out.fill(0)          # touched pages: a copy into never-touched pageable memory faults page by page inside the driver (10x slower)
        self._check(self.lib.mcq_copy_to_host(self.h, out.ctypes.data, int(ptr) + int(offset_bytes), out.nbytes), "mcq_copy_to_host")
        return out

    def sync(self):
        self._check(self.lib.mcq_sync(self.h), "mcq_sync")

    def stream(self):
        return self.lib.mcq_stream(self.h)

    def last_timing_ms(self):
        ms = (ctypes.c_float * 5)()
        self._check(self.lib.mcq_last_timing(self.h, ctypes.byref(ms)), "mcq_last_timing")
        return dict(solve=ms[2], total=ms[4], assemble_sp=ms[0])      # (ms[1], ms[3]: kernels that no longer exist -- include/mcq.h)

    def timing_begin(self):
        """Opens a span of launches timed on the device (mcq_timing_begin)."""
        self._check(self.lib.mcq_timing_begin(self.h), "mcq_timing_begin")

    def timing_end(self):
        """Closes the span: (milliseconds on the device between begin and end, solver launches enqueued in between).  Blocks until the span's last launch is done."""
        ms, cnt = ctypes.c_float(), ctypes.c_int()
        self._check(self.lib.mcq_timing_end(self.h, ctypes.byref(ms), ctypes.byref(cnt)), "mcq_timing_end")
        return float(ms.value), int(cnt.value)

    def workspace_bytes(self):
        return int(self.lib.mcq_workspace_bytes(self.h))

    # ---- the one collective of a multi-GPU job: RCCL's all-gather through the C ABI, on the engine's own stream (include/mcq.h) -----
    COMM_ID_BYTES = 128
    DT_F64, DT_F32, DT_I32 = 0, 1, 2

    def comm_unique_id(self):
        """128 opaque bytes: created by rank 0, shipped to every rank by the launcher's rendezvous."""
        buf = ctypes.create_string_buffer(self.COMM_ID_BYTES)
        rc = self.lib.mcq_comm_unique_id(buf)
        if rc != 0:
            raise EngineError("mcq_comm_unique_id failed: %s" % self.lib.mcq_last_error().decode())
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        if len(unique_id) != self.COMM_ID_BYTES:
            raise ValueError("comm_init: the unique id has %d bytes, not %d" % (len(unique_id), self.COMM_ID_BYTES))
        self._check(self.lib.mcq_comm_init(self.h, int(rank), int(world), bytes(unique_id)), "mcq_comm_init")

    def comm_allgather(self, d_send, d_recv, count, dtype=0):
        """recv [world][count] <- send [count] of every rank (device pointers); asynchronous on the engine's stream."""
        self._check(self.lib.mcq_comm_allgather(self.h, d_send, d_recv, int(count), int(dtype)), "mcq_comm_allgather")

    def comm_wait(self, lag=0):
        """Blocks until the gather enqueued `lag` gathers ago is done; returns its device time in ms (0.0 if there was none)."""
        ms = ctypes.c_float(0.0)
        self._check(self.lib.mcq_comm_wait(self.h, int(lag), ctypes.byref(ms)), "mcq_comm_wait")
        return float(ms.value)

    def comm_world(self):
        r, w = ctypes.c_int(), ctypes.c_int()
        self._check(self.lib.mcq_comm_world(self.h, ctypes.byref(r), ctypes.byref(w)), "mcq_comm_world")
        return r.value, w.value

    def comm_destroy(self):
        self._check(self.lib.mcq_comm_destroy(self.h), "mcq_comm_destroy")


_LIB_FOR_HOST_HELPERS = None


def les_scalings(A, check=True):
    """The N spline scalings out of the dense [4N, 4N] matrix the reference passes as `A` (mcq_les_scalings: one threaded pass over the matrix in
    C, no GPU, no handle) -- the structural check of trajectory_planning_helpers.calc_splines.scalings_from_les_matrix, which stays as its
    numpy statement (and takes any array this entry cannot: other dtypes, non-contiguous views).  Raises RuntimeError with upstream-style
    wording when `A` is not calc_splines' closed-spline system."""
    global _LIB_FOR_HOST_HELPERS
    if not (isinstance(A, np.ndarray) and A.dtype == np.float64 and A.ndim == 2 and A.shape[0] == A.shape[1] and A.shape[0] % 4 == 0
            and A.shape[0] >= 12 and A.flags["C_CONTIGUOUS"]):
        from .trajectory_planning_helpers import calc_splines as _cs
        return _cs.scalings_from_les_matrix(A, check=check)
    if _LIB_FOR_HOST_HELPERS is None:
        _LIB_FOR_HOST_HELPERS = _DEFAULT_ENGINE.lib if _DEFAULT_ENGINE is not None else load_library()
    n = A.shape[0] // 4
    s = np.empty(n)
    if _LIB_FOR_HOST_HELPERS.mcq_les_scalings(A.ctypes.data, n, s.ctypes.data, 1 if check else 0) != 0:
        raise RuntimeError("Spline equation system matrix A does not have the structure of calc_splines' closed-spline "
                           "system (the MI355X engine derives everything from the N spline scalings it encodes and "
                           "cannot use an arbitrary matrix): " + _LIB_FOR_HOST_HELPERS.mcq_last_error().decode())
    return s


_DEFAULT_ENGINE = None


def default_engine():
    """Process-wide engine on the device given by LOCAL_RANK (one process per GPU) or device 0."""
    global _DEFAULT_ENGINE
    if _DEFAULT_ENGINE is None:
        _DEFAULT_ENGINE = Engine(int(os.environ.get("MCQ_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _DEFAULT_ENGINE
