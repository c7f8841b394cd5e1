"""Host-side shim of tph.check_normals_crossing -- boundary [REF helper_funcs_glob/src/prep_track.py:57-59].
Vectorised over the waypoints per neighbour distance (each unordered pair once: the test is symmetric); same predicate as
upstream (restated in oracle/vel_ref.py): collinear normals (|cross| <= 1e-8, numpy.isclose against 0) are skipped, bounds
included.  The batched device form is mcq_normals_crossing_device."""
import numpy as np


def check_normals_crossing(track: np.ndarray, normvec_normalized: np.ndarray, horizon: int = 10) -> bool:
    """True if the normal segments [p - w_l n, p + w_r n] of two points within `horizon` neighbours intersect."""
    n = track.shape[0]
    if horizon >= n:
        raise RuntimeError("Horizon of %i points is too large for a track with %i points, reduce horizon!" % (horizon, n))
    elif horizon >= n / 2:
        print("WARNING: Horizon of %i points makes no sense for a track with %i points, reduce horizon!" % (horizon, n))
    idx = np.arange(n)
    for d in range(1, horizon + 1):
        j = (idx + d) % n
        p0, p1 = track[idx, :2], track[j, :2]
        v0, v1 = normvec_normalized[idx], normvec_normalized[j]
        cross = v0[:, 0] * v1[:, 1] - v0[:, 1] * v1[:, 0]
        ok = np.abs(cross) > 1e-8
        det = np.where(ok, -cross, 1.0)              # p0 + l0 v0 = p1 + l1 v1
        rhs = p1 - p0
        l0 = (-rhs[:, 0] * v1[:, 1] + rhs[:, 1] * v1[:, 0]) / det
        l1 = (v0[:, 0] * rhs[:, 1] - v0[:, 1] * rhs[:, 0]) / det
        hit = ok & (l0 >= -track[idx, 3]) & (l0 <= track[idx, 2]) & (l1 >= -track[j, 3]) & (l1 <= track[j, 2])
        if np.any(hit):
            return True
    return False
