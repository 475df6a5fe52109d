"""Rotated 3-D IoU / GIoU (next row f-2).  CPU: the oracle against closed-form answers (shapely is absent, so the
reference itself cannot be run: parity unpinned, see oracle/iou_oracle.py).  GPU: the HIP kernel against the oracle."""
import math

import numpy as np
import pytest

from oracle import iou_oracle as IO


def box(x=0, y=0, z=0, o=0, l=2, w=1, h=1):
    return np.array([x, y, z, o, l, w, h], float)


def test_oracle_known_answers():
    a = box()
    assert abs(IO.iou3d(a, a) - 2.0 / (2.0 + 1e-5)) < 1e-12                      # identical boxes
    assert abs(IO.giou3d(a, a) - 1.0) < 1e-12
    b = box(x=1.0)                                                                # half overlap along the length
    assert abs(IO.intersection_area(a, b) - 1.0) < 1e-12
    assert abs(IO.iou3d(a, b) - 1.0 / (3.0 + 1e-5)) < 1e-12
    assert abs(IO.giou3d(a, b) - (1.0 / 3.0 - (3.0 - 3.0) / 3.0)) < 1e-12        # hull 3x1x1 = union
    c = box(x=5.0)                                                                # disjoint
    assert IO.iou3d(a, c) == 0.0
    assert abs(IO.giou3d(a, c) - (0.0 - (7.0 - 4.0) / 7.0)) < 1e-12              # hull 7x1, union 4
    d = box(z=0.5)                                                                # half overlap in height
    assert abs(IO.iou3d(a, d) - 1.0 / (3.0 + 1e-5)) < 1e-12
    e = box(z=3.0)                                                                # stacked, no height overlap
    assert IO.iou3d(a, e) == 0.0
    s1, s2 = box(l=2, w=2), box(l=2, w=2, o=math.pi / 4)                          # square vs 45 deg square: octagon
    assert abs(IO.intersection_area(s1, s2) - 8.0 * (math.sqrt(2) - 1.0)) < 1e-12
    big, small = box(l=4, w=4), box(x=0.5, y=-0.3, o=0.7, l=1, w=0.5)             # containment
    assert abs(IO.intersection_area(big, small) - 0.5) < 1e-12
    assert abs(IO.intersection_area(small, big) - 0.5) < 1e-12
    # symmetry and rotation invariance on random boxes
    rng = np.random.default_rng(0)
    for _ in range(50):
        p, q = rng.normal(size=7), rng.normal(size=7)
        p[4:] = np.abs(p[4:]) + 0.2
        q[4:] = np.abs(q[4:]) + 0.2
        assert abs(IO.intersection_area(p, q) - IO.intersection_area(q, p)) < 1e-10
        r = p.copy()
        r[3] += math.pi                                                           # a box equals itself turned by 180 deg
        assert abs(IO.iou3d(p, q) - IO.iou3d(r, q)) < 1e-10


def _random_boxes(rng, n):
    b = np.zeros((n, 7))
    b[:, :2] = rng.uniform(-6, 6, size=(n, 2))
    b[:, 2] = rng.normal(0, 0.5, size=n)
    b[:, 3] = rng.uniform(-math.pi, math.pi, size=n)
    b[:, 4] = rng.uniform(1.0, 5.0, size=n)
    b[:, 5] = rng.uniform(0.5, 2.5, size=n)
    b[:, 6] = rng.uniform(0.5, 2.0, size=n)
    return b


@pytest.mark.gpu
@pytest.mark.parametrize("asso", ["iou", "giou"])
def test_hip_iou_matrix_vs_oracle(asso):
    from shasta_amd import association as A
    rng = np.random.default_rng(1)
    dets, trks = _random_boxes(rng, 37), _random_boxes(rng, 29)
    trks[0] = dets[0]                     # identical
    trks[1] = dets[1] + [0, 0, 0, math.pi / 2, 0, 0, 0]
    trks[2] = dets[2] * [1, 1, 1, 1, 0.3, 0.3, 1]   # contained
    trks[3] = dets[3] + [50, 0, 0, 0, 0, 0, 0]      # far away
    trks[4] = dets[4] + [dets[4][4], 0, 0, 0, 0, 0, 0]  # touching-ish
    got = A.compute_iou_distance(list(dets), list(trks), asso)
    ref = IO.distance_matrix(dets, trks, asso)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    assert got.shape == (37, 29)
    # the association surface consumes it like the reference does
    m, ud, ut = A.associate_dets_to_tracks(list(dets), list(trks), "bipartite", asso, dist_threshold=0.9)
    assert any(int(a) == 0 and int(b) == 0 for a, b in m)


@pytest.mark.gpu
def test_hip_iou_matrix_large_and_empty():
    from shasta_amd import association as A
    rng = np.random.default_rng(2)
    dets, trks = _random_boxes(rng, 500), _random_boxes(rng, 500)
    got = A.compute_iou_distance(list(dets), list(trks), "iou")
    sub = rng.integers(0, 500, size=(200, 2))
    for i, j in sub:
        assert abs(got[i, j] - (1 - IO.iou3d(dets[i], trks[j]))) < 1e-9
    assert got.min() >= -1e-12 and got.max() <= 1.0 + 1e-12
    assert A.compute_iou_distance([], list(trks), "iou").shape == (0, 500)
