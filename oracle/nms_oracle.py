"""CPU oracle (TEST INFRASTRUCTURE) for the rotated BEV NMS of det3d/ops/iou3d_nms (nms_gpu: src/iou3d_nms.cpp:100-143,
nms_kernel src/iou3d_nms_kernel.cu:267-311, iou_bev :226-233): float64 numpy, boxes [x, y, z, dx, dy, dz, heading] sorted by
descending score.  PARITY UNPINNED against the reference for the same reason as oracle/iou_oracle.py (the C++/CUDA sources
need CUDA headers and cannot be built here; the reference's float32 overlap routine is restated as an exact convex clip);
pinned by closed-form known answers in tests/test_nms.py.  Only tests/ may import this module."""
import numpy as np

from oracle import iou_oracle as I


def bev_corners(b):
    cx, cy, hx, hy, a = float(b[0]), float(b[1]), 0.5 * float(b[3]), 0.5 * float(b[4]), float(b[6])
    c, s = np.cos(a), np.sin(a)
    u = np.array([[-hx, -hy], [hx, -hy], [hx, hy], [-hx, hy]])
    return np.stack([cx + u[:, 0] * c - u[:, 1] * s, cy + u[:, 0] * s + u[:, 1] * c], axis=1)


def iou_bev(a, b):
    ov = I.poly_area(I.clip_convex(bev_corners(a), bev_corners(b)))
    sa, sb = float(a[3]) * float(a[4]), float(b[3]) * float(b[4])
    return ov / max(sa + sb - ov, 1e-8)


def iou_matrix(boxes):
    n = len(boxes)
    m = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            m[i, j] = m[j, i] = iou_bev(boxes[i], boxes[j])
    return m


def nms_sorted(boxes, thresh, iou=None):
    """Greedy suppression in the given (score) order; returns kept indices and the IoU matrix used."""
    iou = iou_matrix(boxes) if iou is None else iou
    removed = np.zeros(len(boxes), bool)
    keep = []
    for i in range(len(boxes)):
        if removed[i]:
            continue
        keep.append(i)
        removed |= iou[i] > thresh  # bits of earlier boxes are never consulted again
    return np.array(keep, np.int64), iou


# --------------------------------------------------------------------------------------------------------------------------
# The reference's OWN overlap routine, restated in numpy float32 (box_overlap, det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-225):
# a second, independent algorithm next to the convex clip above.  Edge x edge intersections (:64-96, with the bounding-box
# pre-test :43-49 and the strict straddle test), plus every corner of one box that lies inside the other WITH A MARGIN OF 1e-2
# (check_in_box2d :51-61 - the routine's one quirk: the area can exceed the exact intersection for nearly touching boxes), sorted
# by angle around their mean (:98-100, bubble sort on atan2) and summed as a fan of triangles from the first point (:218-223).
# --------------------------------------------------------------------------------------------------------------------------
_f = np.float32


def _cross3(p1, p2, p0):
    return (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1])


def _ref_intersection(p1, p0, q1, q0):
    if not (min(p0[0], p1[0]) <= max(q0[0], q1[0]) and min(q0[0], q1[0]) <= max(p0[0], p1[0]) and
            min(p0[1], p1[1]) <= max(q0[1], q1[1]) and min(q0[1], q1[1]) <= max(p0[1], p1[1])):
        return None
    s1, s2, s3, s4 = _cross3(q0, p1, p0), _cross3(p1, q1, p0), _cross3(p0, q1, q0), _cross3(q1, p1, q0)
    if not (s1 * s2 > 0 and s3 * s4 > 0):
        return None
    s5 = _cross3(q1, p1, p0)
    if abs(s5 - s1) > _f(1e-8):
        return np.array([(s5 * q0[0] - s1 * q1[0]) / (s5 - s1), (s5 * q0[1] - s1 * q1[1]) / (s5 - s1)], _f)
    a0, b0, c0 = p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]
    a1, b1, c1 = q0[1] - q1[1], q1[0] - q0[0], q0[0] * q1[1] - q1[0] * q0[1]
    D = a0 * b1 - a1 * b0
    return np.array([(b0 * c1 - b1 * c0) / D, (a1 * c0 - a0 * c1) / D], _f)


def _ref_corners(b):
    cx, cy, hx, hy = _f(b[0]), _f(b[1]), _f(b[3]) / _f(2), _f(b[4]) / _f(2)
    c, s = _f(np.cos(_f(b[6]))), _f(np.sin(_f(b[6])))
    pts = np.array([[cx - hx, cy - hy], [cx + hx, cy - hy], [cx + hx, cy + hy], [cx - hx, cy + hy]], _f)
    out = np.empty((4, 2), _f)
    for k in range(4):
        dx, dy = pts[k, 0] - cx, pts[k, 1] - cy
        out[k] = (dx * c + dy * (-s) + cx, dx * s + dy * c + cy)
    return out


def _ref_in_box(b, p, margin):
    c, s = _f(np.cos(-_f(b[6]))), _f(np.sin(-_f(b[6])))
    dx, dy = p[0] - _f(b[0]), p[1] - _f(b[1])
    rx, ry = dx * c + dy * (-s), dx * s + dy * c
    return abs(rx) < _f(b[3]) / _f(2) + _f(margin) and abs(ry) < _f(b[4]) / _f(2) + _f(margin)


def ref_box_overlap(a, b, margin=1e-2):
    """Overlap area as the reference's kernel computes it (float32).  margin=0 gives the routine without its quirk."""
    with np.errstate(all="ignore"):
        ca, cb = _ref_corners(a), _ref_corners(b)
        pts = []
        for i in range(4):
            for j in range(4):
                x = _ref_intersection(ca[(i + 1) % 4], ca[i], cb[(j + 1) % 4], cb[j])
                if x is not None:
                    pts.append(x)
        for k in range(4):
            if _ref_in_box(a, cb[k], margin):
                pts.append(cb[k])
            if _ref_in_box(b, ca[k], margin):
                pts.append(ca[k])
        if not pts:
            return 0.0
        pts = np.array(pts, _f)
        ctr = pts.sum(0, dtype=_f) / _f(len(pts))
        ang = [_f(np.arctan2(p[1] - ctr[1], p[0] - ctr[0])) for p in pts]
        order = list(range(len(pts)))
        for j in range(len(order) - 1):           # the reference's bubble sort (stable for equal angles)
            for i in range(len(order) - j - 1):
                if ang[order[i]] > ang[order[i + 1]]:
                    order[i], order[i + 1] = order[i + 1], order[i]
        pts = pts[order]
        area = _f(0)
        for k in range(len(pts) - 1):
            u, v = pts[k] - pts[0], pts[k + 1] - pts[0]
            area += u[0] * v[1] - u[1] * v[0]
        return float(abs(area) / _f(2))


def near_boundary(a, b, tol):
    """True when a corner of one box lies within `tol` of the other box's boundary (where the reference's margin can act)."""
    for p, q in ((a, b), (b, a)):
        c, s = np.cos(-float(q[6])), np.sin(-float(q[6]))
        for x, y in bev_corners(p):
            dx, dy = x - float(q[0]), y - float(q[1])
            rx, ry = abs(dx * c - dy * s) - float(q[3]) / 2, abs(dx * s + dy * c) - float(q[4]) / 2
            if abs(max(rx, ry)) < tol or (abs(rx) < tol and ry < tol) or (abs(ry) < tol and rx < tol):
                return True
    return False


def iou_normal(a, b):
    """iou3d_nms_kernel.cu:313-323 in float32, operation for operation."""
    a, b = np.asarray(a, _f), np.asarray(b, _f)
    two = _f(2)
    left, right = max(a[0] - a[3] / two, b[0] - b[3] / two), min(a[0] + a[3] / two, b[0] + b[3] / two)
    top, bottom = max(a[1] - a[4] / two, b[1] - b[4] / two), min(a[1] + a[4] / two, b[1] + b[4] / two)
    w, h = max(right - left, _f(0)), max(bottom - top, _f(0))
    inter = _f(w * h)
    return _f(inter / max(_f(_f(a[3] * a[4]) + _f(b[3] * b[4])) - inter, _f(1e-8)))


def nms_normal_sorted(boxes, thresh):
    n = len(boxes)
    iou = np.zeros((n, n), _f)
    for i in range(n):
        for j in range(i + 1, n):
            iou[i, j] = iou[j, i] = iou_normal(boxes[i], boxes[j])
    keep, _ = nms_sorted(boxes, _f(thresh), iou)
    return keep, iou


def boxes_iou3d(boxes_a, boxes_b, overlap=None):
    """iou3d_nms_utils.py:35-72 on det3d-convention boxes [x, y, z, w, l, h, theta]: float64 with the exact overlap by default."""
    def pc(b):
        b = np.asarray(b, np.float64)[:, [0, 1, 2, 4, 3, 5, 6]].copy()
        b[:, 6] = -b[:, 6] - np.pi / 2
        return b
    A, B = pc(boxes_a), pc(boxes_b)
    out = np.zeros((len(A), len(B)))
    for i, a in enumerate(A):
        for j, b in enumerate(B):
            ov = I.poly_area(I.clip_convex(bev_corners(a), bev_corners(b))) if overlap is None else overlap(a, b)
            oh = max(min(a[2] + a[5] / 2, b[2] + b[5] / 2) - max(a[2] - a[5] / 2, b[2] - b[5] / 2), 0.0)
            o3 = ov * oh
            out[i, j] = o3 / max(a[3] * a[4] * a[5] + b[3] * b[4] * b[5] - o3, 1e-6)
    return out
