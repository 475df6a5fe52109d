"""The call surface of det3d/ops/iou3d_nms/iou3d_nms_utils.py on the device: `boxes_iou_bev` (:13-27),
`to_pcdet` (:29-33), `boxes_iou3d_gpu` (:35-72), `nms_gpu` (:74-89), `nms_normal_gpu` (:93-106), and
det3d/core/bbox/box_torch_ops.py:248-276 (`rotate_nms_pcdet`), computed by csrc/nms.hip through `shasta_boxes_bev_f32`,
`shasta_nms_rotated_f32` and `shasta_nms_normal_f32`.  Device tensors only (no CPU path); nothing is copied to the host: the
reference reduces the suppression mask on the CPU (src/iou3d_nms.cpp:113-140), here the reduction is a second kernel."""
import math

import torch

from . import hip


def _bev_matrix(boxes_a, boxes_b, mode):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    if not (boxes_a.is_cuda and boxes_b.is_cuda):
        raise hip.ShastaHipError("the IoU matrices need device tensors; there is no CPU path")
    lib = hip.load()
    a, b = boxes_a.float().contiguous(), boxes_b.float().contiguous()
    out = torch.zeros(a.shape[0], b.shape[0], device=a.device)
    hip.check(lib.shasta_boxes_bev_f32(hip.ptr(a) if a.shape[0] else None, a.shape[0], hip.ptr(b) if b.shape[0] else None, b.shape[0], mode,
                                       hip.ptr(out) if out.numel() else None, hip.stream_ptr()), "shasta_boxes_bev_f32")
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    """boxes_a (N, 7), boxes_b (M, 7) [x, y, z, dx, dy, dz, heading] -> (N, M) rotated BEV IoU (iou3d_nms_utils.py:13-27)."""
    return _bev_matrix(boxes_a, boxes_b, 1)


def boxes_overlap_bev(boxes_a, boxes_b):
    """(N, M) overlap areas of the rotated footprints (`iou3d_nms_cuda.boxes_overlap_bev_gpu`, used at iou3d_nms_utils.py:57-58)."""
    return _bev_matrix(boxes_a, boxes_b, 0)


def to_pcdet(boxes):
    """det3d's [x, y, z, w, l, h, theta] rows to pcdet's [x, y, z, dx, dy, dz, heading = -theta - pi/2] (iou3d_nms_utils.py:29-33)."""
    boxes = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    boxes[:, -1] = -boxes[:, -1] - math.pi / 2
    return boxes


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N, 7), (M, 7) boxes in det3d's convention -> (N, M) 3-D IoU (iou3d_nms_utils.py:35-72): converted with to_pcdet, then BEV
    overlap x height overlap over the union volume (clamped at 1e-6), all inside one kernel."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return _bev_matrix(to_pcdet(boxes_a), to_pcdet(boxes_b), 2)


def _nms_sorted(boxes_sorted, thresh, entry="shasta_nms_rotated_f32"):
    lib = hip.load()
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.shasta_nms_workspace_bytes(n)
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=dev)
    hip.check(getattr(lib, entry)(hip.ptr(boxes_sorted) if n else None, n, float(thresh), hip.ptr(ws), ws_bytes, hip.ptr(keep),
                                  hip.ptr(num), hip.stream_ptr()), entry)
    return keep[:int(num.item())].long()


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """boxes (N,7) [x, y, z, dx, dy, dz, heading], scores (N,) -> (indices of the kept boxes in score order, None)."""
    assert boxes.shape[1] == 7
    if not boxes.is_cuda:
        raise hip.ShastaHipError("nms_gpu needs device tensors; there is no CPU path")
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].float().contiguous()
    return order[_nms_sorted(b, thresh)].contiguous(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:93-106: NMS on the axis-aligned BEV footprints (heading ignored); (kept indices in score order, None)."""
    assert boxes.shape[1] == 7
    if not boxes.is_cuda:
        raise hip.ShastaHipError("nms_normal_gpu needs device tensors; there is no CPU path")
    order = scores.sort(0, descending=True)[1]
    b = boxes[order].float().contiguous()
    return order[_nms_sorted(b, thresh, "shasta_nms_normal_f32")].contiguous(), None


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """boxes (N,7) [x, y, z, l, w, h, theta] in det3d's convention (box_torch_ops.py:248-276): swapped to pcdet's
    [x, y, z, dx=w.., heading = -theta - pi/2] exactly like the reference, then NMS."""
    b = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    b[:, -1] = -b[:, -1] - math.pi / 2
    selected, _ = nms_gpu(b, scores, thresh, pre_maxsize=pre_maxsize)
    if post_max_size is not None:
        selected = selected[:post_max_size]
    return selected
