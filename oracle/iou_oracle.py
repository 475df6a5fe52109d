"""CPU oracle (TEST INFRASTRUCTURE) for the rotated 3-D IoU / GIoU distance matrices of `mot_3d.association`
(asso='iou' / 'giou'): a float64 numpy restatement of
  mot_3d/utils/geometry.py:161-176 (iou3d), :208-231 (giou3d), :234-237 (PolyArea2D),
  mot_3d/data_protos/bbox.py:70-84 (box2corners2d), mot_3d/association.py:108-120 (compute_iou_distance).

PARITY UNPINNED against the reference: the reference computes the polygon intersection with shapely (GEOS), which is
not installed here and not vendored; the rotated-IoU C++/CUDA code under det3d/ops/iou3d_nms needs CUDA headers and is
unbuildable in this image.  The intersection of two convex quadrilaterals is restated with Sutherland-Hodgman clipping
(any exact algorithm yields the same area up to rounding); the convex hull uses scipy.spatial.ConvexHull exactly like the
reference.  The oracle is pinned by closed-form known answers instead (tests/test_iou.py).
Only tests/ may import this module.
"""
import numpy as np
from scipy.spatial import ConvexHull


def corners2d(b):
    """b = [x, y, z, o, l, w, h] -> (4, 2) corners pc0..pc3 (bbox.py:70-84)."""
    x, y, _, o, l, w, _ = (float(v) for v in b[:7])
    c, s = np.cos(o), np.sin(o)
    p0 = np.array([x + c * l / 2 + s * w / 2, y + s * l / 2 - c * w / 2])
    p1 = np.array([x + c * l / 2 - s * w / 2, y + s * l / 2 + c * w / 2])
    ctr = np.array([x, y])
    return np.stack([p0, p1, 2 * ctr - p0, 2 * ctr - p1])


def poly_area(p):
    if len(p) < 3:
        return 0.0
    q = np.roll(p, -1, axis=0)
    return abs(float(np.sum(p[:, 0] * q[:, 1] - p[:, 1] * q[:, 0]))) * 0.5


def _signed_area(p):
    q = np.roll(p, -1, axis=0)
    return float(np.sum(p[:, 0] * q[:, 1] - p[:, 1] * q[:, 0])) * 0.5


def clip_convex(subject, clip):
    """Sutherland-Hodgman: part of convex polygon `subject` inside convex polygon `clip` (vertex arrays)."""
    sgn = 1.0 if _signed_area(clip) >= 0 else -1.0
    out = [np.asarray(v, float) for v in subject]
    n = len(clip)
    for i in range(n):
        a, b = clip[i], clip[(i + 1) % n]
        if not out:
            break
        inp, out = out, []
        e = b - a

        def side(p):
            return sgn * (e[0] * (p[1] - a[1]) - e[1] * (p[0] - a[0]))

        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                out.append(p + t * (q - p))
    return np.array(out) if out else np.zeros((0, 2))


def intersection_area(a, b):
    return poly_area(clip_convex(corners2d(a), corners2d(b)))


def _heights(a, b):
    za, zb, ha, hb = float(a[2]), float(b[2]), float(a[6]), float(b[6])
    d1 = (za + ha / 2) - (zb - hb / 2)
    d2 = (zb + hb / 2) - (za - ha / 2)
    return max(0.0, min(d1, d2)), max(d1, d2)


def iou3d(a, b):
    """geometry.py:161-176, second return value."""
    ov = intersection_area(a, b)
    oh, _ = _heights(a, b)
    vol = ov * oh
    union = float(a[5]) * float(a[4]) * float(a[6]) + float(b[5]) * float(b[4]) * float(b[6]) - vol
    return vol / (union + 1e-5)


def giou3d(a, b):
    """geometry.py:208-231."""
    oh, uh = _heights(a, b)
    I = intersection_area(a, b) * oh
    U = float(a[5]) * float(a[4]) * float(a[6]) + float(b[5]) * float(b[4]) * float(b[6]) - I
    pts = np.vstack([corners2d(a), corners2d(b)])
    hull = ConvexHull(pts)
    C = poly_area(pts[hull.vertices]) * uh
    return I / U - (C - U) / C


def distance_matrix(dets, tracks, asso):
    """association.py:108-120: 1 - iou matrix, rows = detections, columns = tracks."""
    fn = iou3d if asso == "iou" else giou3d
    m = np.zeros((len(dets), len(tracks)))
    for i, d in enumerate(dets):
        for j, t in enumerate(tracks):
            m[i, j] = fn(d, t)
    return 1 - m
