"""Decode of the affinity matrices into per-frame detection lists, restating the inline loop of
tools/nusc_shasta/eval.py:112-181 (== validate.py:53-122) as functions, so the consumer side of the hot path is callable
and testable.  One device->host copy per frame (the reference does one .item() per element)."""
import numpy as np
import torch


def decode_frame(matched1, matched2, cls_det_boxes, prev_cls_det_boxes, token, time_lag):
    """matched1 (N, N+2), matched2 (N+2, N) for ONE frame pair (tensor or array).  The box dict lists are mutated exactly
    like the reference does (FN boxes are moved forward by time_lag*velocity and re-tokened, flags added).
    Returns (annos, dead_prev_idx, keep_det_idx)."""
    m1 = matched1.detach().cpu().numpy() if torch.is_tensor(matched1) else np.asarray(matched1)
    m2 = matched2.detach().cpu().numpy() if torch.is_tensor(matched2) else np.asarray(matched2)
    n_prev, n_cur = len(prev_cls_det_boxes), len(cls_det_boxes)
    annos, fn_annos, dead_prev = [], [], []
    if n_prev > 0:
        keep_prev = []
        A = np.concatenate([m1[:n_prev, :n_cur], m1[:n_prev, -2:]], axis=1)
        col_dead, col_fn = A.shape[1] - 2, A.shape[1] - 1
        best = A.argmax(axis=1)
        for n in range(n_prev):
            k = int(best[n])
            val = float(A[n, k])
            if val > 0.5 and k == col_dead:
                dead_prev.append(n)
                continue
            if val > 0.5 and k == col_fn:
                box = prev_cls_det_boxes[n]
                box["translation"][:2] = [t + time_lag * v for t, v in zip(box["translation"][:2], box["velocity"])]
                box["FN"] = True
                box["token"] = token
                box["ref_detection_score"] = 1 - float(A[n, -2])
                fn_annos.append(box)
                continue
            keep_prev.append(n)
        Bm = np.concatenate([m2[keep_prev, :n_cur], m2[-2:, :n_cur]], axis=0)
    else:
        Bm = m2[-2:, :n_cur]
    keep_dets = []
    if n_cur > 0:
        row_fp, row_newborn = Bm.shape[0] - 1, Bm.shape[0] - 2
        best = Bm.argmax(axis=0)
        for k in range(n_cur):
            n = int(best[k])
            val = float(Bm[n, k])
            if val > 0.7 and n == row_fp:
                continue
            if val > 0.5 and n == row_newborn:
                cls_det_boxes[k]["newborn"] = True
            cls_det_boxes[k]["ref_detection_score"] = 1 - float(Bm[-1, k])
            keep_dets.append(k)
            annos.append(cls_det_boxes[k])
    annos.extend(fn_annos)
    return annos, dead_prev, keep_dets


class AffinityDecoder:
    """Accumulates frames like eval.py's main loop: results dict token -> annos, plus the dead-track post-pass."""

    def __init__(self):
        self.results = {}
        self.dead_tracker = {}

    def add(self, matched1, matched2, processed_batch, b=0):
        token = processed_batch["metadata"][b]["token"]
        self.dead_tracker.setdefault(token, {"dead_idx": [], "keep_idx": []})
        cls = processed_batch["cls_det_boxes"][b]
        prev_cls = processed_batch["prev_cls_det_boxes"][b]
        time_lag = float(processed_batch["prev_det_boxes"][b, 0, 9]) if len(prev_cls) else 0.0
        annos, dead_prev, keep = decode_frame(matched1[b], matched2[b], cls, prev_cls, token, time_lag)
        if len(prev_cls):
            prev_token = processed_batch["prev_metadata"][b]["token"]
            self.dead_tracker.setdefault(prev_token, {"dead_idx": [], "keep_idx": []})
            self.dead_tracker[prev_token]["dead_idx"].extend(dead_prev)
        if len(cls):
            self.dead_tracker[token]["keep_idx"] = keep
        self.results[token] = annos
        return annos

    def finalize(self):
        for token, annos in self.results.items():
            info = self.dead_tracker[token]
            for i in info["dead_idx"]:
                if i in info["keep_idx"]:
                    annos[info["keep_idx"].index(i)]["dead"] = True
        return {"results": self.results,
                "meta": {"use_camera": False, "use_lidar": True, "use_radar": False, "use_map": False,
                         "use_external": False}}
