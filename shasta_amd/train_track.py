"""Caller-side contract of the hot path: `example_to_device` and `track_batch_processor`, mirroring
det3d/torchie/apis/train_track.py:29-74 and :109-130 (same keys, same casts, same return tuples)."""
import torch

_FLOAT_KEYS = {
    "voxels", "bev_map", "prev_bev_map", "bev_feature", "prev_bev_feature", "coordinates", "num_points",
    "num_voxels", "prev_voxels", "prev_coordinates", "prev_num_points", "prev_num_voxels", "cyv_voxels",
    "cyv_num_voxels", "cyv_coordinates", "cyv_num_points", "det_boxes", "prev_det_boxes", "det_boxes_emb",
    "prev_det_boxes_emb", "gt", "gt_matched", "gt_newborn", "pos_wt",
}
_LIST_KEYS = {"anchors", "anchors_mask", "reg_targets", "reg_weights", "labels", "points", "prev_points"}


def example_to_device(example, device=None, non_blocking=False) -> dict:
    """train_track.py:29-74: tensors in the float list go to the GPU as fp32 (integer coordinates and counts
    included), list-valued keys element-wise, everything else is passed through untouched."""
    assert device is not None or torch.cuda.is_available()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = {}
    for k, v in example.items():
        if k in _LIST_KEYS:
            out[k] = [r.to(dev, non_blocking=non_blocking) for r in v]
        elif k in _FLOAT_KEYS:
            out[k] = torch.as_tensor(v).to(dev).to(dtype=torch.float)
        elif k == "calib":
            out[k] = {k1: torch.tensor(v1).to(dev) for k1, v1 in v.items()}
        elif k in ("num_det_boxes", "num_prev_det_boxes"):
            out[k] = torch.tensor(v).to(dev).to(dtype=torch.float)
        else:
            out[k] = v
    return out


def track_batch_processor(model, data, train_mode=True, **kwargs):
    """train_track.py:109-130: returns (matched1, matched2, gt) in train mode, else (matched1, matched2, example)."""
    device = torch.device("cuda", kwargs["local_rank"]) if "local_rank" in kwargs else None
    example = example_to_device(data, device, non_blocking=False)
    out = model(example, train_mode=train_mode)
    if train_mode:
        return out[0], out[1], out[-1]["gt"]
    return out
