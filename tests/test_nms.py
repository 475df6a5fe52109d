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
