"""Input formats of the affinity path: the per-frame detection files the reference's dataset reads and the
(max_obj, 11) box rows / (max_obj+2)^2 ground-truth matrix it builds from them, restated from
det3d/datasets/nuscenes/nuscenes.py:198-349 (`NuScenesDataset.get_sensor_data`, the part before the LiDAR pipeline).

On-disk layout (written by the reference's preprocessing/):
  <det_path>/<token>.json        list of 13-float rows [t(3), wlh(3), quaternion wxyz(4), vxy(2), score]   (LiDAR frame)
  <cls_info_path>/<token>.json   list of nuScenes detection dicts ('detection_name', 'detection_score', ...)
  <frame_info>.json              {token: {'prev': token | '', 'timestamp': us, 'prev_timestamp': us, ...}}
  <labels_path>/<token>.npz      matched (N_prev, K+2) one-hot rows [K current detections | dead track | unused], newborn (K,)

Box row: [x, y, z, w, l, h, yaw, vx, vy, dt, score] (nuscenes.py:230-233).  The random sub-sampling (more than max_obj
detections, dead-track / false-positive ratios) draws from `rng` in the same order as the reference draws from the global
`random` module, so `random.seed(s)` reproduces the reference's sample exactly (tests/test_frames.py).
"""
import json
import math
import os
import random as _random

import numpy as np


def quaternion_yaw(q_wxyz):
    """nuscenes.py:35-49: yaw of the rotated x axis, v = R(q) [1,0,0], atan2(v_y, v_x).  pyquaternion normalises the quaternion
    before building the rotation matrix; so does this."""
    w, x, y, z = (float(v) for v in q_wxyz)
    n = math.sqrt(w * w + x * x + y * y + z * z)
    if n > 0.0:
        w, x, y, z = w / n, x / n, y / n, z / n
    return math.atan2(2.0 * (x * y + w * z), 1.0 - 2.0 * (y * y + z * z))


def det_rows(boxes13, cls_info, det_type, time_diff, max_objects, rng=_random):
    """nuscenes.py:220-246 / 263-294 for one frame.  Returns (rows (max_objects, 11) float64 zero padded, keep: indices into
    the file's detection list, kept class dicts, number of valid rows).  With no detection left, `keep` is
    range(max_objects), as in the reference."""
    rows = np.zeros((max_objects, 11))
    keep = list(range(max_objects))
    kept_cls = []
    if len(boxes13) == 0:
        return rows, keep, kept_cls, 0
    keep, dets = [], []
    for i, (b, ci) in enumerate(zip(boxes13, cls_info)):
        if det_type is not None and ci["detection_name"] not in det_type:
            continue
        dets.append(np.concatenate((np.asarray(b[:3], dtype=float), np.asarray(b[3:6], dtype=float),
                                    np.array([quaternion_yaw(b[6:10])]), np.asarray(b[10:12], dtype=float),
                                    np.array([time_diff]), np.array([ci["detection_score"]], dtype=float))))
        kept_cls.append(ci)
        keep.append(i)
    n = 0
    if len(dets) > 0:
        if len(dets) > max_objects:
            pick = rng.sample(range(len(dets)), max_objects)
            pick.sort()
            dets = [dets[i] for i in pick]
            kept_cls = [kept_cls[i] for i in pick]
            keep = [keep[i] for i in pick]
        n = len(dets)
        rows[:n, :] = np.array(dets)
    return rows, keep, kept_cls, n


def assemble_gt(matched, newborn, prev_keep, keep, has_prev, max_objects, fp_ratio, dead_trk_ratio, rng=_random):
    """nuscenes.py:297-349: the (max_objects+2)^2 target [tracks | newborn | false positive] x [detections | dead | false
    negative] with dead tracks and false positives sub-sampled to a ratio of the true positives.
    Returns (gt, num_prev_det_boxes or None (unchanged), num_det_boxes)."""
    N = max_objects
    gt = np.zeros((N + 2, N + 2))
    num_prev = None
    if has_prev:
        P = len(prev_keep)
        gt[:P, :] = 0
        temp = matched[prev_keep][:, keep]
        gt[:P, :len(keep)] = temp
        gt[:P, -2] = matched[prev_keep, -2]
        gt[:P, -1] = 1 - gt[:P, :].sum(axis=1)
        dead_trk, fn = gt[:P, -2], gt[:P, -1]
        prev_tp = gt[:P, :-2].sum(axis=1) + fn
        prev_tp_idx = list(np.nonzero(prev_tp == 1)[0])
        dead_trk_idx = list(np.nonzero(dead_trk == 1)[0])
        rng.shuffle(dead_trk_idx)
        keep_dead = dead_trk_idx[:int(dead_trk_ratio * prev_tp.sum())]
        tpk = keep_dead + prev_tp_idx
        tpk.sort()
        num_prev = len(tpk)
        gt[:len(tpk), :] = gt[tpk, :]
        gt[len(tpk):-2, :] = np.zeros((N - len(tpk), N + 2))
    K = len(keep)
    gt[-2, :K] = newborn[keep]
    fp = 1 - gt[:, :K].sum(axis=0)
    gt[-1, :K] = fp
    tp = gt[:-1, :K].sum(axis=0)
    tp_idx = list(np.nonzero(tp == 1)[0])
    fp_idx = list(np.nonzero(fp == 1)[0])
    rng.shuffle(fp_idx)
    keep_fp = fp_idx[:int(fp_ratio * tp.sum())]
    tk = keep_fp + tp_idx
    tk.sort()
    gt[:, :len(tk)] = gt[:, tk]
    gt[:, len(tk):-2] = np.zeros((N + 2, N - len(tk)))
    return gt, num_prev, len(tk)


class FramePairs:
    """The detection side of `NuScenesDataset` (constructor names follow nuscenes.py:57-110): `load(token)` returns what
    `get_sensor_data` puts into `info` before the LiDAR pipeline runs - det_boxes, prev_det_boxes, their class dicts and
    counts, and in training mode the ground-truth matrix."""

    def __init__(self, det_path, cls_info_path, frame_info_path, labels_path=None, det_type=None, max_objects=500,
                 fp_ratio=1.0, dead_trk_ratio=1.0, test_mode=False, rng=_random):
        self.det_path, self.cls_info_path, self.labels_path = det_path, cls_info_path, labels_path
        self.det_type, self.max_objects = det_type, max_objects
        self.fp_ratio, self.dead_trk_ratio, self.test_mode, self.rng = fp_ratio, dead_trk_ratio, test_mode, rng
        with open(frame_info_path) as f:
            self.frame_info = json.load(f)

    def _read(self, token):
        with open(os.path.join(self.det_path, token + ".json")) as f:
            boxes = json.load(f)
        with open(os.path.join(self.cls_info_path, token + ".json")) as f:
            cls_info = json.load(f)
        return boxes, cls_info

    def load(self, token, known_tokens=None):
        fi = self.frame_info[token]
        prev_token = fi["prev"]
        if known_tokens is not None and prev_token not in known_tokens:
            prev_token = ""  # nuscenes.py:202-204: previous frame not part of this split
        N = self.max_objects
        time_diff = 1e-6 * fi["timestamp"] - 1e-6 * fi["prev_timestamp"] if "prev_timestamp" in fi else 0.0
        out = dict(token=token, prev_token=prev_token)
        prev_rows, prev_keep, prev_cls, n_prev = np.zeros((N, 11)), list(range(N)), [], 0
        if prev_token != "":
            b, c = self._read(prev_token)
            prev_rows, prev_keep, prev_cls, n_prev = det_rows(b, c, self.det_type, time_diff, N, self.rng)
        b, c = self._read(token)
        rows, keep, cls, n = det_rows(b, c, self.det_type, time_diff, N, self.rng)
        out.update(prev_det_boxes=prev_rows, prev_cls_det_boxes=prev_cls, num_prev_det_boxes=n_prev, det_boxes=rows,
                   cls_det_boxes=cls, num_det_boxes=n)
        if not self.test_mode:
            lab = np.load(os.path.join(self.labels_path, token + ".npz"), allow_pickle=True)
            gt, np_, nd = assemble_gt(lab["matched"], lab["newborn"], prev_keep, keep, prev_token != "", N, self.fp_ratio,
                                      self.dead_trk_ratio, self.rng)
            out["gt"] = gt
            if np_ is not None:
                out["num_prev_det_boxes"] = np_
            out["num_det_boxes"] = nd
        return out


def parse_frame(det_path, cls_info_path, token):
    """Both files of a frame, read and parsed once: (number of detections, (n, 11) float64 rows [x, y, z, w, l, h, yaw, vx, vy, 0, score]
    (box rows with column 9 - the time difference - left open), the class dicts, {class name: indices of its detections})."""
    with open(os.path.join(det_path, token + ".json")) as f:
        boxes = json.load(f)
    with open(os.path.join(cls_info_path, token + ".json")) as f:
        cls_info = json.load(f)
    by_class = {}
    for i, ci in enumerate(cls_info):
        by_class.setdefault(ci["detection_name"], []).append(i)
    base = np.zeros((len(boxes), 11))
    if len(boxes):
        # quaternion_yaw for all rows at once: the same float64 operations in the same order (products, sums, sqrt and quotients
        # are exactly rounded in numpy as in Python); atan2 stays math.atan2 (numpy's may be a vector-library version)
        b = np.array(boxes, dtype=np.float64).reshape(len(boxes), -1)
        w, x, y, z = b[:, 6], b[:, 7], b[:, 8], b[:, 9]
        n = np.sqrt(w * w + x * x + y * y + z * z)
        n = np.where(n > 0.0, n, 1.0)
        w, x, y, z = w / n, x / n, y / n, z / n
        sn, cs = 2.0 * (x * y + w * z), 1.0 - 2.0 * (y * y + z * z)
        base[:, 0:6] = b[:, 0:6]
        base[:, 6] = [math.atan2(p, q) for p, q in zip(sn.tolist(), cs.tolist())]
        base[:, 7:9] = b[:, 10:12]
        base[:, 10] = [float(ci["detection_score"]) for ci in cls_info]
    return len(boxes), base, cls_info, by_class


class SharedFrames:
    """`FramePairs.load` for SEVERAL detection classes over the same frames: every per-frame file is read and parsed once (the
    reference's per-class eval runs re-read both files of a frame twice per class, as current and as previous frame), the quaternion ->
    yaw conversion is done once per detection, and the rows of a class are assembled from the cached values.  `load(name, token)`
    returns exactly what `FramePairs(det_type=[name], max_objects=max_objects[name], test_mode=True).load(token)` returns - the same
    float64 rows, fresh class dicts (the decode mutates them) - as tests/test_frames.py checks on the synthetic split."""

    def __init__(self, det_path, cls_info_path, frame_info_path, max_objects, rng=_random, cache=256):
        self.det_path, self.cls_info_path, self.max_objects, self.rng, self.cache = det_path, cls_info_path, dict(max_objects), rng, cache
        with open(frame_info_path) as f:
            self.frame_info = json.load(f)
        self._parsed = {}

    def _frame(self, token):
        fr = self._parsed.get(token)
        if fr is None:
            fr = self.adopt(token, parse_frame(self.det_path, self.cls_info_path, token))
        return fr

    def adopt(self, token, parsed):
        """Puts a parsed frame (parse_frame's tuple) into the cache."""
        if len(self._parsed) >= self.cache:
            self._parsed.pop(next(iter(self._parsed)))
        fr = self._parsed[token] = tuple(parsed) + ({},)
        return fr

    def _rows(self, name, token, time_diff):
        N = self.max_objects[name]
        nbox, base, cls_info, by_class, _ = self._frame(token)
        rows = np.zeros((N, 11))
        if nbox == 0:
            return rows, list(range(N)), [], 0
        keep = list(by_class.get(name, ()))
        if len(keep) > N:
            pick = self.rng.sample(range(len(keep)), N)
            pick.sort()
            keep = [keep[i] for i in pick]
        if keep:
            r = base[keep]
            r[:, 9] = time_diff
            rows[:len(keep)] = r
        # fresh dicts: decode.decode_frame writes flags / scores into them and moves `translation` of a propagated box in place
        kept_cls = [dict(cls_info[i], translation=list(cls_info[i]["translation"])) for i in keep]
        return rows, keep, kept_cls, len(keep)

    def _class_rows(self, name, token):
        """(indices of the class's detections in the frame, their (k, 11) float64 rows with column 9 = 0), cached with the frame."""
        fr = self._frame(token)
        hit = fr[4].get(name)
        if hit is None:
            keep = fr[3].get(name, [])
            hit = fr[4][name] = (keep, fr[1][keep])
        return hit

    def load_run(self, name, run, share_prev=False, out=None):
        """Every frame pair of a run of consecutive frames ([(token, prev token or "")], pipeline._scene_runs) for one class, as the stacked
        batch `collate_pairs([load(name, t) for t in run])` gives (same fp32 rows, same class dicts), without building each frame twice
        (as current frame and as the next one's previous frame): the rows of a (frame, class) are assembled once and written straight into
        the (n, max_obj, 11) fp32 stacks.  share_prev: the previous frame's class dicts are the cached ones, NOT fresh copies - for a
        consumer that copies before it writes (decode.decode_frame_from_flags(copy_fn=True)); the current frame's dicts are always fresh.
        out: (det, prev) arrays to fill instead of fresh ones.  Returns None when a frame holds more than max_obj detections of the class (random sub-sampling: use `load`)."""
        N, n = self.max_objects[name], len(run)
        if out is None:
            det, prev = np.zeros((n, N, 11), np.float32), np.zeros((n, N, 11), np.float32)
        else:  # the caller's (n, N, 11) fp32 arrays (pinned staging memory of the chain): filled in place
            det, prev = out
            det[...] = 0.0
            prev[...] = 0.0
        out = dict(det_boxes=det, prev_det_boxes=prev, num_det_boxes=[], num_prev_det_boxes=[], cls_det_boxes=[], prev_cls_det_boxes=[],
                   metadata=[], prev_metadata=[])
        for i, (token, prev_token) in enumerate(run):
            fi = self.frame_info[token]
            time_diff = 1e-6 * fi["timestamp"] - 1e-6 * fi["prev_timestamp"] if "prev_timestamp" in fi else 0.0
            keep, rows = self._class_rows(name, token)
            k = len(keep)
            if k > N:
                return None
            cls_info = self._frame(token)[2]
            if k:
                det[i, :k] = rows
                det[i, :k, 9] = time_diff
            out["cls_det_boxes"].append([dict(cls_info[j], translation=list(cls_info[j]["translation"])) for j in keep])
            out["num_det_boxes"].append(k)
            kp, prev_cls = 0, []
            if prev_token != "":
                pkeep, prows = self._class_rows(name, prev_token)
                kp = len(pkeep)
                if kp > N:
                    return None
                pinfo = self._frame(prev_token)[2]
                if kp:
                    prev[i, :kp] = prows
                    prev[i, :kp, 9] = time_diff
                prev_cls = [pinfo[j] for j in pkeep] if share_prev else [dict(pinfo[j], translation=list(pinfo[j]["translation"])) for j in pkeep]
            out["prev_cls_det_boxes"].append(prev_cls)
            out["num_prev_det_boxes"].append(kp)
            out["metadata"].append(dict(token=token))
            out["prev_metadata"].append(dict(token=prev_token or token))
        return out

    def load(self, name, token, known_tokens=None):
        fi = self.frame_info[token]
        prev_token = fi["prev"]
        if known_tokens is not None and prev_token not in known_tokens:
            prev_token = ""
        N = self.max_objects[name]
        time_diff = 1e-6 * fi["timestamp"] - 1e-6 * fi["prev_timestamp"] if "prev_timestamp" in fi else 0.0
        prev_rows, prev_cls, n_prev = np.zeros((N, 11)), [], 0
        if prev_token != "":
            prev_rows, _, prev_cls, n_prev = self._rows(name, prev_token, time_diff)
        rows, _, cls, n = self._rows(name, token, time_diff)
        return dict(token=token, prev_token=prev_token, prev_det_boxes=prev_rows, prev_cls_det_boxes=prev_cls, num_prev_det_boxes=n_prev,
                    det_boxes=rows, cls_det_boxes=cls, num_det_boxes=n)


def collate_pairs(samples, device=None):
    """Stack loaded frame pairs into the batch the model consumes (det3d/torchie/parallel/collate.py keeps these keys as
    stacked float tensors; example_to_device casts to fp32): det_boxes / prev_det_boxes (B, max_obj, 11) fp32, gt
    (B, max_obj+2, max_obj+2) fp32 when present.  Any number of frames and detection classes can be stacked: every pair is an
    independent batch row of the affinity forward (B >> 1 is what the device path is built for)."""
    import torch
    batch = dict(
        det_boxes=torch.from_numpy(np.stack([s["det_boxes"] for s in samples]).astype(np.float32)),
        prev_det_boxes=torch.from_numpy(np.stack([s["prev_det_boxes"] for s in samples]).astype(np.float32)),
        num_det_boxes=[s["num_det_boxes"] for s in samples], num_prev_det_boxes=[s["num_prev_det_boxes"] for s in samples],
        cls_det_boxes=[s["cls_det_boxes"] for s in samples], prev_cls_det_boxes=[s["prev_cls_det_boxes"] for s in samples],
        metadata=[dict(token=s["token"]) for s in samples],
        # decode.AffinityDecoder.add reads it (eval.py:118); a frame without predecessor carries its own token there, because
        # the reference then runs the current frame's LiDAR a second time as "previous" data (nuscenes.py:399-406)
        prev_metadata=[dict(token=s["prev_token"] or s["token"]) for s in samples])
    if all("gt" in s for s in samples):
        batch["gt"] = torch.from_numpy(np.stack([s["gt"] for s in samples]).astype(np.float32))
    if device is not None:
        for k in ("det_boxes", "prev_det_boxes", "gt"):
            if k in batch:
                batch[k] = batch[k].pin_memory().to(device, non_blocking=True) if device.type == "cuda" else batch[k].to(device)
    return batch
