"""PubTracker with the N x M centre-distance matrix and the greedy assignment on the device.

Same constructor, `reset()` and `step_centertrack(results, time_lag)` contract as tools/nusc_shasta/pub_tracker.py:35-210
(the detection dicts are annotated in place and returned, track ids, ages, `active` counters, the `newborn` / `dead`
suppression rules and the confidence refinement behave identically); the distance matrix, the class / velocity gate and
`greedy_assignment` (track_utils.py:3-14) run in one kernel launch (csrc/track.hip).  `step_batch` advances many independent
scenes with ONE launch: the greedy loop is sequential inside a scene, so the device only pays off across scenes
(tools/nusc_shasta/eval.py:251-259 runs the scenes one after the other).  The Hungarian option keeps scipy's solver on the
host, fed with the device-computed matrix, like the reference."""
import copy

import numpy as np
import torch

from . import hip

NUSCENES_TRACKING_NAMES = ["bicycle", "bus", "car", "motorcycle", "pedestrian", "trailer", "truck"]

# 99.9 percentile of the l2 velocity error per class / 0.5 s (pub_tracker.py:21-31)
NUSCENE_CLS_VELOCITY_ERROR = {"car": 2, "truck": 2, "bus": 4, "trailer": 2, "pedestrian": 0.75, "motorcycle": 2, "bicycle": 1.5}


def center_greedy_device(problems, device=None, want_dist=True):
    """problems: list of (dets (N,2) f32, tracks (M,2) f32, det_cat (N,) i32, trk_cat (M,) i32, max_diff (N,) f32) numpy
    tuples with N, M >= 1.  Returns a list of (dist (N,M) float64 numpy or None, matched_indices (K,2) int32 numpy)."""
    lib = hip.load()
    device = device or torch.device("cuda", torch.cuda.current_device())
    S = len(problems)
    if S == 0:
        return []
    Nmax = max(p[0].shape[0] for p in problems)
    Mmax = max(p[1].shape[0] for p in problems)
    dxy, txy = np.zeros((S, Nmax, 2), np.float32), np.zeros((S, Mmax, 2), np.float32)
    dc, tc = np.zeros((S, Nmax), np.int32), np.zeros((S, Mmax), np.int32)
    md = np.zeros((S, Nmax), np.float32)
    n, m = np.zeros(S, np.int32), np.zeros(S, np.int32)
    for s, (d, t, a, b, g) in enumerate(problems):
        n[s], m[s] = d.shape[0], t.shape[0]
        dxy[s, :n[s]], txy[s, :m[s]], dc[s, :n[s]], tc[s, :m[s]], md[s, :n[s]] = d, t, a, b, g
    dev = [torch.from_numpy(x).to(device) for x in (dxy, txy, dc, tc, md, n, m)]
    dist = torch.empty(S, Nmax, Mmax, dtype=torch.float64, device=device) if want_dist else None
    match = torch.empty(S, Nmax, dtype=torch.int32, device=device)
    hip.check(lib.shasta_center_greedy_f32(*[hip.ptr(x) for x in dev], S, Nmax, Mmax, hip.ptr(dist), hip.ptr(match), hip.stream_ptr()),
              "shasta_center_greedy_f32")
    match_h = match.cpu().numpy()
    dist_h = dist.cpu().numpy() if want_dist else None
    out = []
    for s in range(S):
        mi = match_h[s, :n[s]]
        rows = np.nonzero(mi >= 0)[0]
        pairs = np.stack([rows, mi[rows]], axis=1).astype(np.int32).reshape(-1, 2)
        out.append((dist_h[s, :n[s], :m[s]].copy() if want_dist else None, pairs))
    return out


class PubTracker(object):
    def __init__(self, hungarian=False, max_age=0, refine_confidence=False, alpha=0.5, beta=0.5, verbose=False):
        self.hungarian = hungarian
        self.max_age = max_age
        self.refine_confidence = refine_confidence
        self.alpha = alpha
        self.beta = beta
        self.NUSCENE_CLS_VELOCITY_ERROR = NUSCENE_CLS_VELOCITY_ERROR
        if verbose:  # the reference prints these two lines unconditionally
            print("Use hungarian: {}".format(hungarian))
            print(self.NUSCENE_CLS_VELOCITY_ERROR)
        self.reset()

    def reset(self):
        self.id_count = 0
        self.tracks = []

    # ---- the three phases of pub_tracker.py:55-210 -------------------------------------------------------------------
    def _prepare(self, results, time_lag):
        """:55-93: filter to tracking classes, annotate ct / tracking / label_preds, build the arrays of the distance step.
        Returns None when the frame has no detection at all (the caller clears the tracks)."""
        if len(results) == 0:
            return None
        temp = []
        for det in results:
            if det["detection_name"] not in NUSCENES_TRACKING_NAMES:
                continue
            det["ct"] = np.array(det["translation"][:2])
            det["tracking"] = np.array(det["velocity"][:2]) * -1 * time_lag
            det["label_preds"] = NUSCENES_TRACKING_NAMES.index(det["detection_name"])
            temp.append(det)
        results = temp
        if "tracking" in results[0]:  # IndexError on a frame without any tracking-class detection, like the reference
            dets = np.array([det["ct"] + det["tracking"].astype(np.float32) for det in results], np.float32)
        else:
            dets = np.array([det["ct"] for det in results], np.float32)
        item_cat = np.array([item["label_preds"] for item in results], np.int32)
        track_cat = np.array([track["label_preds"] for track in self.tracks], np.int32)
        max_diff = np.array([self.NUSCENE_CLS_VELOCITY_ERROR[box["detection_name"]] for box in results], np.float32)
        tracks = np.array([pre_det["ct"] for pre_det in self.tracks], np.float32)
        return results, dets, tracks, item_cat, track_cat, max_diff

    def _finish(self, results, dets, tracks, dist, matched_indices):
        """:118-210: unmatched bookkeeping, ids, ages, confidence refinement."""
        unmatched_dets = [d for d in range(dets.shape[0]) if not (d in matched_indices[:, 0])]
        unmatched_tracks = [d for d in range(tracks.shape[0]) if not (d in matched_indices[:, 1])]
        if self.hungarian:
            matches = []
            for m in matched_indices:
                if dist[m[0], m[1]] > 1e16:
                    unmatched_dets.append(m[0])
                else:
                    matches.append(m)
            matches = np.array(matches).reshape(-1, 2)
        else:
            matches = matched_indices
        ret = []
        for m in matches:
            track = results[m[0]]
            track["tracking_id"] = self.tracks[m[1]]["tracking_id"]
            if self.refine_confidence:
                prev_track_conf = self.tracks[m[1]]["ref_detection_score"]
                tp_prob = track["ref_detection_score"]
                det_conf = track["detection_score"]
                track["ref_detection_score"] = (tp_prob > self.alpha) * self.beta * det_conf + (1 - self.beta) * prev_track_conf
            track["age"] = 1
            track["active"] = self.tracks[m[1]]["active"] + 1
            ret.append(track)
        for i in unmatched_dets:
            track = results[i]
            if len(tracks) > 0:
                if "newborn" not in track.keys() and (dist[i, :] <= self.NUSCENE_CLS_VELOCITY_ERROR[track["detection_name"]]).sum():
                    continue
            self.id_count += 1
            track["tracking_id"] = self.id_count
            track["ref_detection_score"] = track["detection_score"]
            track["age"] = 1
            track["active"] = 1
            ret.append(track)
        for i in unmatched_tracks:
            track = self.tracks[i]
            if "dead" in track.keys() and (dist[:, i] <= self.NUSCENE_CLS_VELOCITY_ERROR[track["detection_name"]]).sum():
                continue
            if track["age"] < self.max_age:
                track["age"] += 1
                track["active"] = 0
                ct = track["ct"]
                if "tracking" in track:
                    offset = track["tracking"] * -1  # move forward
                    track["ct"] = ct + offset
                ret.append(track)
        self.tracks = ret
        return ret

    def _host_assign(self, dist):
        from scipy.optimize import linear_sum_assignment
        d = dist.copy()
        d[d > 1e18] = 1e18
        row_ind, col_ind = linear_sum_assignment(copy.deepcopy(d))
        return d, np.concatenate((row_ind.reshape(-1, 1), col_ind.reshape(-1, 1)), axis=-1)

    def step_centertrack(self, results, time_lag):
        return step_batch([self], [results], [time_lag])[0]


def step_batch(trackers, results_list, time_lags):
    """One tracker step for many independent scenes; the distance / greedy work of all of them is one kernel launch."""
    prepared, problems, where = [], [], []
    for k, (trk, results, lag) in enumerate(zip(trackers, results_list, time_lags)):
        p = trk._prepare(results, lag)
        prepared.append(p)
        if p is not None and len(p[2]) > 0:  # not the first frame of the scene
            where.append(k)
            problems.append((p[1], p[2], p[3], p[4], p[5]))
    solved = dict(zip(where, center_greedy_device(problems))) if problems else {}
    outs = []
    for k, trk in enumerate(trackers):
        p = prepared[k]
        if p is None:
            trk.tracks = []
            outs.append([])
            continue
        results, dets, tracks = p[0], p[1], p[2]
        if k in solved:
            dist, matched = solved[k]
            if trk.hungarian:
                dist, matched = trk._host_assign(dist)
        else:
            assert len(trk.tracks) == 0
            dist, matched = None, np.array([], np.int32).reshape(-1, 2)
        outs.append(trk._finish(results, dets, tracks, dist, matched))
    return outs
