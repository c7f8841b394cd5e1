"""
Deterministic synthetic reference tracks for BASELINE.json configs 3 and 5:

  perturbed 2:1 oval, perimeter ~6000 m, N = 2000 waypoints equidistant in arclength (3.0 m step), centreline offset
  along its normal by 8 m sin(2 pi 7 s/P) + 3 m sin(2 pi 23 s/P) so that the optimum has a partial active set;
  widths per batch item b:  w_r, w_l = 5.0 + 1.5 u,  u ~ U(-1, 1) i.i.d. per waypoint, moving-average(15) low-pass,
  rng = default_rng(1234 + b).

Host-side numpy only; produces exactly the arrays the reference's prep_track would hand to opt_min_curv
([x, y, w_tr_right, w_tr_left] rows, unit normals pointing right, spline scalings).
"""
import numpy as np

from .trajectory_planning_helpers import calc_splines as _cs


_BASE = {}


def _base_oval(n, perimeter, ratio, fine=200):
    """Finely sampled 2:1 ellipse scaled to the perimeter, its arclength stations and right-pointing normals (the part
    of oval_centreline that does not depend on the track; cached)."""
    key = (int(n), float(perimeter), float(ratio), int(fine))
    if key not in _BASE:
        m = fine * n
        th = np.linspace(0.0, 2.0 * np.pi, m, endpoint=False)
        # clockwise or counter-clockwise does not matter to the QP; use counter-clockwise
        ex, ey = ratio * np.cos(th), np.sin(th)
        seg = np.hypot(np.diff(np.append(ex, ex[0])), np.diff(np.append(ey, ey[0])))
        scale = perimeter / float(np.sum(seg))
        ex, ey, seg = ex * scale, ey * scale, seg * scale
        s = np.concatenate(([0.0], np.cumsum(seg)[:-1]))
        tx, ty = np.gradient(ex), np.gradient(ey)
        tn = np.hypot(tx, ty)
        if len(_BASE) > 3:
            _BASE.clear()
        _BASE[key] = (ex, ey, s, ty / tn, -tx / tn)
    return _BASE[key]


def oval_centreline(n=2000, perimeter=6000.0, ratio=2.0, amp1=8.0, k1=7, amp2=3.0, k2=23, centre_seed=None, fine=200):
    """fine: samples of the underlying curve per waypoint (200 for the shared centreline of configs 3; oval_batch uses 20
    for per-track centrelines -- config 5 generates 8192 of them per rank -- which moves a waypoint by < 1e-3 m)."""
    ex, ey, s, nx, ny = _base_oval(n, perimeter, ratio, fine)
    off = amp1 * np.sin(2.0 * np.pi * k1 * s / perimeter) + amp2 * np.sin(2.0 * np.pi * k2 * s / perimeter)
    if centre_seed is not None:
        rng = np.random.default_rng(centre_seed)
        for k in rng.integers(3, 40, size=4):
            off = off + rng.uniform(0.5, 2.5) * np.sin(2.0 * np.pi * k * s / perimeter + rng.uniform(0, 2 * np.pi))
    px, py = ex + off * nx, ey + off * ny
    seg2 = np.hypot(np.diff(np.append(px, px[0])), np.diff(np.append(py, py[0])))
    s2 = np.concatenate(([0.0], np.cumsum(seg2)))
    tgt = np.linspace(0.0, s2[-1], n, endpoint=False)
    x = np.interp(tgt, s2, np.append(px, px[0]))
    y = np.interp(tgt, s2, np.append(py, py[0]))
    return np.column_stack((x, y))


def widths(n, b, base=5.0, amp=1.5, window=15):
    rng = np.random.default_rng(1234 + int(b))
    u = rng.uniform(-1.0, 1.0, size=(n, 2))
    ker = np.ones(window) / window
    pad = window // 2
    out = np.empty_like(u)
    for c in range(2):
        ext = np.concatenate((u[-pad:, c], u[:, c], u[:pad, c]))
        out[:, c] = np.convolve(ext, ker, mode="valid")
    return base + amp * out


def prepared_track(xy):
    """normals + spline scalings as the reference's prep_track produces them [REF prep_track.py:48-51]."""
    path_cl = np.vstack((xy, xy[0]))
    s = _cs.spline_scalings(path_cl)
    _, b, _, _ = _cs.closed_spline_coeffs(xy, s)
    nv = np.stack((b[:, 1], -b[:, 0]), axis=1)
    nv /= np.sqrt(np.sum(nv ** 2, axis=1))[:, None]
    return nv, s


def oval_batch(batch, n=2000, first=0, perturb_centreline=False):
    """Returns reftrack [B, n, 4], normvec [B, n, 2], scaling [B, n] (float64, C-contiguous)."""
    ref = np.empty((batch, n, 4))
    nv = np.empty((batch, n, 2))
    sc = np.empty((batch, n))
    xy0 = nv0 = s0 = None
    if not perturb_centreline:
        xy0 = oval_centreline(n)
        nv0, s0 = prepared_track(xy0)
    for i in range(batch):
        b = first + i
        if perturb_centreline:
            xy0 = oval_centreline(n, centre_seed=b, fine=20)
            nv0, s0 = prepared_track(xy0)
        ref[i, :, :2] = xy0
        ref[i, :, 2:] = widths(n, b)
        nv[i] = nv0
        sc[i] = s0
    return ref, nv, sc
