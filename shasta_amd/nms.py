"""Rotated BEV NMS on the device: the call surface of det3d/ops/iou3d_nms/iou3d_nms_utils.py:74-89
(`nms_gpu`) and det3d/core/bbox/box_torch_ops.py:248-276 (`rotate_nms_pcdet`), computed by csrc/nms.hip through
`shasta_nms_rotated_f32`.  Device tensors only (no CPU path); nothing is copied to the host: the reference reduces the
suppression mask on the CPU (src/iou3d_nms.cpp:113-140), here the reduction is a second kernel."""
import math

import torch

from . import hip


def _nms_sorted(boxes_sorted, thresh):
    lib = hip.load()
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.shasta_nms_workspace_bytes(n)
    ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=dev)
    hip.check(lib.shasta_nms_rotated_f32(hip.ptr(boxes_sorted) if n else None, n, float(thresh), hip.ptr(ws), ws_bytes, hip.ptr(keep),
                                         hip.ptr(num), hip.stream_ptr()), "shasta_nms_rotated_f32")
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


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """boxes (N,7) [x, y, z, l, w, h, theta] in det3d's convention (box_torch_ops.py:248-276): swapped to pcdet's
    [x, y, z, dx=w.., heading = -theta - pi/2] exactly like the reference, then NMS."""
    b = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    b[:, -1] = -b[:, -1] - math.pi / 2
    selected, _ = nms_gpu(b, scores, thresh, pre_maxsize=pre_maxsize)
    if post_max_size is not None:
        selected = selected[:post_max_size]
    return selected
