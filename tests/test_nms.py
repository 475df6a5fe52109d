"""Rotated BEV NMS: oracle known answers on CPU, device kernel against the oracle on the GPU."""
import math

import numpy as np
import pytest
import torch

from oracle import nms_oracle as NO


def test_oracle_iou_known_answers():
    a = [0, 0, 0, 4, 2, 1, 0.0]
    assert abs(NO.iou_bev(a, a) - 1.0) < 1e-12
    assert abs(NO.iou_bev(a, [2, 0, 0, 4, 2, 1, 0.0]) - (4.0 / 12.0)) < 1e-12            # half overlap along x
    assert abs(NO.iou_bev(a, [0, 0, 0, 2, 4, 1, math.pi / 2]) - 1.0) < 1e-12             # same rectangle, rotated description
    assert abs(NO.iou_bev(a, [0, 0, 0, 4, 2, 1, math.pi / 2]) - (4.0 / 12.0)) < 1e-12    # cross: 2x2 overlap
    assert NO.iou_bev(a, [10, 0, 0, 4, 2, 1, 0.3]) == 0.0
    sq = [0, 0, 0, 2, 2, 1, 0.0]
    oct_area = 8 * (math.sqrt(2) - 1)                                                     # square ∩ square rotated by 45°
    assert abs(NO.iou_bev(sq, [0, 0, 0, 2, 2, 1, math.pi / 4]) - oct_area / (8 - oct_area)) < 1e-12


def test_oracle_nms_known_answer():
    boxes = np.array([[0, 0, 0, 4, 2, 1, 0], [0.2, 0, 0, 4, 2, 1, 0], [10, 0, 0, 4, 2, 1, 0], [10.1, 0.1, 0, 4, 2, 1, 0.05],
                      [0, 5, 0, 4, 2, 1, 1.0]], float)
    keep, _ = NO.nms_sorted(boxes, 0.5)
    assert list(keep) == [0, 2, 4]
    keep, _ = NO.nms_sorted(boxes, 0.99)
    assert list(keep) == [0, 1, 2, 3, 4]


def _random_boxes(rng, n, spread):
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 3:6] = rng.uniform(0.5, 5.0, (n, 3))
    b[:, 6] = rng.uniform(-math.pi, math.pi, n)
    return b


@pytest.mark.gpu
@pytest.mark.parametrize("n,spread,thresh", [(1, 1, 0.5), (5, 2, 0.1), (64, 8, 0.3), (65, 8, 0.3), (200, 12, 0.2), (700, 25, 0.45)])
def test_device_nms_matches_oracle(n, spread, thresh):
    from shasta_amd import nms
    rng = np.random.default_rng(n)
    boxes = _random_boxes(rng, n, spread)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    dev = torch.device("cuda:0")
    sel, _ = nms.nms_gpu(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), thresh)
    order = np.argsort(-scores, kind="stable")
    keep, iou = NO.nms_sorted(boxes[order].astype(np.float64), thresh)
    want = order[keep]
    got = sel.cpu().numpy()
    assert not (np.abs(iou[np.triu_indices(n, 1)] - thresh) < 1e-6).any()  # no decision within rounding of the threshold
    assert got.dtype == np.int64 and np.array_equal(got, want)


@pytest.mark.gpu
def test_device_nms_edges_and_pcdet_wrapper():
    from shasta_amd import nms
    dev = torch.device("cuda:0")
    sel, _ = nms.nms_gpu(torch.zeros(0, 7, device=dev), torch.zeros(0, device=dev), 0.5)
    assert sel.numel() == 0
    # identical boxes: only the best-scoring one survives; pre_maxsize / post_max_size are honoured
    b = torch.tensor([[0, 0, 0, 2, 4, 1, 0.3]] * 6 + [[20, 0, 0, 2, 4, 1, 0.0], [40, 0, 0, 2, 4, 1, 0.0]], device=dev)
    s = torch.tensor([0.1, 0.9, 0.3, 0.2, 0.5, 0.4, 0.8, 0.7], device=dev)
    sel, _ = nms.nms_gpu(b, s, 0.5)
    assert sel.tolist() == [1, 6, 7]
    assert nms.nms_gpu(b, s, 0.5, pre_maxsize=2)[0].tolist() == [1, 6]
    assert nms.rotate_nms_pcdet(b.clone(), s, 0.5, post_max_size=2).tolist() == [1, 6]
    assert torch.equal(b[0], torch.tensor([0, 0, 0, 2, 4, 1, 0.3], device=dev))  # caller's boxes untouched


@pytest.mark.gpu
def test_device_nms_many_boxes_analytic_chain():
    """6000 unit squares on a line, 0.6 apart: only neighbours overlap (IoU = 0.4 / 1.6 = 0.25), so the greedy result is
    computable without an N x N matrix.  Exercises more than one 64-word slot per lane of the reduction kernel."""
    from shasta_amd import nms
    n = 6000
    rng = np.random.default_rng(0)
    boxes = np.zeros((n, 7), np.float32)
    boxes[:, 0] = 0.6 * np.arange(n)
    boxes[:, 3:6] = 1.0
    scores = rng.permutation(n).astype(np.float32)
    dev = torch.device("cuda:0")
    sel, _ = nms.nms_gpu(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.2)
    removed = np.zeros(n, bool)
    want = []
    for k in np.argsort(-scores, kind="stable"):
        if removed[k]:
            continue
        want.append(k)
        for j in (k - 1, k + 1):
            if 0 <= j < n:
                removed[j] = True
    assert sel.cpu().numpy().tolist() == want
    sel2, _ = nms.nms_gpu(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), 0.3)  # above every IoU
    assert sel2.numel() == n


# ---- the rest of the reference module's surface: boxes_iou_bev / boxes_iou3d_gpu / nms_normal_gpu (parity unpinned: see oracle header) ----

def test_two_independent_overlap_algorithms_agree_parity_unpinned():
    """oracle/nms_oracle.py holds two routes to the overlap area of two rotated rectangles: the exact convex clip the device kernel
    also uses, and a float32 restatement of the REFERENCE's own routine (edge intersections + contained corners, angular sort, triangle
    fan; iou3d_nms_kernel.cu:104-225).  They agree to float32 rounding wherever the reference's 1e-2 corner margin cannot act, and
    differ by at most a margin-wide strip where it can - which is the documented difference between this library and the reference."""
    rng = np.random.default_rng(0)
    boxes = _random_boxes(rng, 45, 6)
    far, near = 0.0, 0.0
    n_near = 0
    for i in range(len(boxes)):
        for j in range(i + 1, len(boxes)):
            exact = NO.I.poly_area(NO.I.clip_convex(NO.bev_corners(boxes[i]), NO.bev_corners(boxes[j])))
            ref = NO.ref_box_overlap(boxes[i], boxes[j])
            if NO.near_boundary(boxes[i], boxes[j], 3e-2):
                near, n_near = max(near, abs(exact - ref)), n_near + 1
            else:
                far = max(far, abs(exact - ref))
    assert far < 2e-5 and near < 0.06 and n_near > 5   # the strip: margin 1e-2 x an edge of at most 5
    a = [0, 0, 0, 4, 2, 1, 0.0]
    assert abs(NO.ref_box_overlap(a, a) - 8.0) < 1e-5 and abs(NO.ref_box_overlap(a, [0, 0, 0, 4, 2, 1, math.pi / 2]) - 4.0) < 1e-5
    # the quirk itself: a 5 mm gap between two boxes is an overlap for the reference, none for the exact clip
    b = [4.005, 0, 0, 4, 2, 1, 0.0]
    assert NO.I.poly_area(NO.I.clip_convex(NO.bev_corners(a), NO.bev_corners(b))) == 0.0 and NO.ref_box_overlap(a, b) > 0.0


def test_oracle_iou_normal_and_iou3d_known_answers():
    a, b = [0, 0, 0, 4, 2, 1, 0.7], [2, 0, 0, 4, 2, 1, -1.1]           # headings are ignored by the axis-aligned form
    assert abs(float(NO.iou_normal(a, b)) - 4.0 / 12.0) < 1e-6
    assert float(NO.iou_normal(a, [10, 0, 0, 1, 1, 1, 0])) == 0.0
    keep, _ = NO.nms_normal_sorted(np.array([a, b, [0.1, 0, 0, 4, 2, 1, 0]], np.float32), 0.5)
    assert list(keep) == [0, 1]
    # det3d convention [x, y, z, w, l, h, theta]: identical boxes -> 1; half height overlap of identical footprints -> (V/2) / (1.5 V)
    d = np.array([[1, 2, 0, 2, 4, 2, 0.3]], np.float64)
    assert abs(NO.boxes_iou3d(d, d)[0, 0] - 1.0) < 1e-9
    up = d.copy()
    up[0, 2] += 1.0
    assert abs(NO.boxes_iou3d(d, up)[0, 0] - 1.0 / 3.0) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("na,nb,spread", [(1, 1, 1), (7, 130, 4), (200, 65, 10)])
def test_device_iou_matrices_match_both_oracle_routes_parity_unpinned(na, nb, spread):
    from shasta_amd import nms
    rng = np.random.default_rng(na * 1000 + nb)
    A, Bx = _random_boxes(rng, na, spread), _random_boxes(rng, nb, spread)
    A[:, 2], Bx[:, 2] = rng.normal(0, 1, na), rng.normal(0, 1, nb)
    dev = torch.device("cuda:0")
    ta, tb = torch.from_numpy(A).to(dev), torch.from_numpy(Bx).to(dev)
    ov = nms.boxes_overlap_bev(ta, tb).cpu().numpy()
    iou = nms.boxes_iou_bev(ta, tb).cpu().numpy()
    assert ov.shape == (na, nb) and ov.dtype == np.float32
    sub = [(i, j) for i in range(min(na, 25)) for j in range(min(nb, 25))]
    for i, j in sub:
        exact = NO.I.poly_area(NO.I.clip_convex(NO.bev_corners(A[i]), NO.bev_corners(Bx[j])))
        assert abs(ov[i, j] - exact) <= 1e-6 * max(1.0, exact)
        assert abs(iou[i, j] - NO.iou_bev(A[i], Bx[j])) <= 2e-6
        if not NO.near_boundary(A[i], Bx[j], 3e-2):   # the reference's own routine, where its corner margin cannot act
            assert abs(ov[i, j] - NO.ref_box_overlap(A[i], Bx[j])) <= 2e-5 * max(1.0, exact)
    # 3-D IoU on det3d-convention rows (the wrapper swaps w / l and maps theta like to_pcdet)
    da, db = A[:, [0, 1, 2, 4, 3, 5, 6]].copy(), Bx[:, [0, 1, 2, 4, 3, 5, 6]].copy()
    got = nms.boxes_iou3d_gpu(torch.from_numpy(da).to(dev), torch.from_numpy(db).to(dev)).cpu().numpy()
    want = NO.boxes_iou3d(da[:25], db[:25])
    assert np.abs(got[:25, :25] - want).max() <= 5e-6
    # the float64 association matrix of mot_3d (shasta_iou3d_distance_f64, another entry point) sees the same overlaps
    assert np.array_equal(nms.boxes_iou_bev(ta, ta).cpu().numpy().diagonal() > 0.999, np.ones(na, bool))
    assert nms.boxes_iou_bev(torch.zeros(0, 7, device=dev), tb).shape == (0, nb)


@pytest.mark.gpu
@pytest.mark.parametrize("n,spread,thresh", [(1, 1, 0.5), (64, 6, 0.3), (65, 6, 0.1), (300, 14, 0.25)])
def test_device_nms_normal_equals_the_reference_arithmetic(n, spread, thresh):
    """nms_normal_gpu: the axis-aligned IoU is plain fp32 in the reference's operation order, so the keep list equals the float32
    restatement of iou3d_nms_kernel.cu:313-372 exactly - no tolerance."""
    from shasta_amd import nms
    rng = np.random.default_rng(n + 7)
    boxes = _random_boxes(rng, n, spread)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    dev = torch.device("cuda:0")
    sel, none = nms.nms_normal_gpu(torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), thresh)
    order = np.argsort(-scores, kind="stable")
    keep, _ = NO.nms_normal_sorted(boxes[order], thresh)
    assert none is None and np.array_equal(sel.cpu().numpy(), order[keep])
