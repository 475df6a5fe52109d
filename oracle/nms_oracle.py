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
