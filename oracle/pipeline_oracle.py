"""CPU oracle of the inference chain of BASELINE configs 2-4 -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's flow restated one frame at a time, exactly as its CLIs run it (batch_size=1, one tracker that is reset at every
scene start): tools/nusc_shasta/eval.py:96-193 (forward -> decode loop -> `dead` post-pass -> cp json),
merge_results.py:37-59, pub_test.py:88-162 / eval.py:226-300 (tracker loop -> tracking json).  Built from the pinned oracles:
shasta_oracle.forward_from_bev / decode_frame / mark_dead and tracker_oracle.  The per-frame detection files are read with the
package's loader (shasta_amd.frames, itself pinned on the reference dataset code by tests/test_frames.py).  Only tests import
this."""
import copy
import json

import numpy as np
import torch

from oracle import shasta_oracle as O
from oracle import tracker_oracle as TO

NAMES = TO.NAMES
META = {"use_camera": False, "use_lidar": True, "use_radar": False, "use_map": False, "use_external": False}


def eval_class(weights, name, max_objects, paths, tokens, bev, num_feats=3, num_point=5, known_tokens=None):
    from shasta_amd import frames
    ds = frames.FramePairs(paths["det_path"], paths["cls_info_path"], paths["frame_info_path"], det_type=[name], max_objects=max_objects,
                           test_mode=True)
    known = set(ds.frame_info.keys()) if known_tokens is None else set(known_tokens)
    results, dead_tracker, margins = {}, {}, []
    for token in tokens:
        s = ds.load(token, known_tokens=known)
        det = torch.from_numpy(s["det_boxes"].astype(np.float32))[None]
        prev = torch.from_numpy(s["prev_det_boxes"].astype(np.float32))[None]
        m1, m2 = O.forward_from_bev(weights, bev(token)[None], bev(s["prev_token"] or token)[None], det, prev, num_feats, num_point)
        cls, prev_cls = s["cls_det_boxes"], s["prev_cls_det_boxes"]
        time_lag = float(prev[0, 0, 9]) if len(prev_cls) else 0.0
        margins.append(decision_margin(m1[0].numpy(), m2[0].numpy(), len(prev_cls), len(cls)))
        annos, dead_prev, keep = O.decode_frame(m1[0].numpy(), m2[0].numpy(), cls, prev_cls, token, time_lag)
        dead_tracker.setdefault(token, {"dead_idx": [], "keep_idx": []})
        if len(prev_cls):
            dead_tracker.setdefault(s["prev_token"], {"dead_idx": [], "keep_idx": []})["dead_idx"].extend(dead_prev)
        if len(cls):
            dead_tracker[token]["keep_idx"] = keep
        results[token] = annos
    O.mark_dead(results, dead_tracker)
    return {"results": results, "meta": dict(META)}, min(margins) if margins else 1.0


def decision_margin(m1, m2, n_prev, n_cur):
    """How far the decode decisions of a frame are from flipping under a perturbation of the matrices: the smallest of (a) the
    distance of every deciding maximum from its threshold (0.5 / 0.7, eval.py:137,141,161,163) and (b) the top-2 gap of every
    row / column whose arg-max is consulted.  The pipeline tests require this to exceed the kernels' error on their scenes."""
    m = 1.0
    if n_prev > 0:
        A = np.concatenate([m1[:n_prev, :n_cur], m1[:n_prev, -2:]], axis=1)
        srt = np.sort(A, axis=1)
        top, gap = srt[:, -1], (srt[:, -1] - srt[:, -2]) if A.shape[1] > 1 else np.ones(n_prev)
        m = min(m, float(np.abs(top - 0.5).min()), float(gap[top > 0.45].min()) if (top > 0.45).any() else 1.0)
    if n_cur > 0:
        keep = [n for n in range(n_prev)] if n_prev > 0 else []
        Bm = np.concatenate([m2[keep, :n_cur], m2[-2:, :n_cur]], axis=0) if n_prev > 0 else m2[-2:, :n_cur]
        srt = np.sort(Bm, axis=0)
        top, gap = srt[-1], srt[-1] - srt[-2]
        m = min(m, float(np.abs(top - 0.5).min()), float(np.abs(top - 0.7).min()), float(gap[top > 0.45].min()) if (top > 0.45).any() else 1.0)
    return m


def merge_results(per_class):
    out = {"meta": dict(META), "results": {}}
    for name in NAMES:
        if name in per_class:
            for token, annos in per_class[name]["results"].items():
                out["results"].setdefault(token, []).extend(annos)
    return out


def run_tracking(predictions, frames_meta, max_age=4, hungarian=False):
    """pub_test.py:88-162: one merged tracker walking the frames in file order, reset at every scene start."""
    tracker = TO.PubTrackerMergedOracle(max_age=max_age, hungarian=hungarian)
    annos = {"results": {}, "meta": dict(META)}
    last = None
    for fr in frames_meta:
        if fr["first"]:
            tracker.reset()
            last = fr["timestamp"]
        lag = fr["timestamp"] - last
        last = fr["timestamp"]
        out = tracker.step_centertrack(predictions[fr["token"]], lag)
        annos["results"][fr["token"]] = [
            {"sample_token": fr["token"], "translation": it["translation"], "size": it["size"], "rotation": it["rotation"],
             "velocity": it["velocity"], "tracking_id": str(it["tracking_id"]), "tracking_name": it["detection_name"],
             "tracking_score": it["ref_detection_score"]} for it in out if it["active"] != 0]
    return annos


def run_split(weights_by_class, max_objects_by_class, paths, scenes, bev):
    tokens = [t for _, toks in scenes for t in toks]
    per_class, margin = {}, 1.0
    for name in NAMES:
        if name in weights_by_class:
            per_class[name], mg = eval_class(weights_by_class[name], name, max_objects_by_class[name], paths, tokens, bev, known_tokens=tokens)
            margin = min(margin, mg)
    merged = merge_results(per_class)
    with open(paths["frames_meta_path"]) as f:
        meta = json.load(f)["frames"]
    return per_class, merged, run_tracking(copy.deepcopy(merged["results"]), meta), margin
