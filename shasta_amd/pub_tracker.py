"""PubTracker with the N x M centre-distance matrix and the greedy assignment on the device.

Same constructor, `reset()` and `step_centertrack(results, time_lag)` contract as tools/nusc_shasta/pub_tracker.py:35-210
(the detection dicts are annotated in place and returned, track ids, ages, `active` counters, the `newborn` / `dead`
suppression rules and the confidence refinement behave identically); the distance matrix, the class / velocity gate and
`greedy_assignment` (track_utils.py:3-14) run in one kernel launch (csrc/track.hip).  `step_batch` advances many independent
scenes with ONE launch: the greedy loop is sequential inside a scene, so the device only pays off across scenes
(tools/nusc_shasta/eval.py:251-259 runs the scenes one after the other).  The Hungarian option keeps scipy's solver on the
host, fed with the device-computed matrix, like the reference."""
import copy

import contextlib

import numpy as np
import torch

from . import hip

NUSCENES_TRACKING_NAMES = ["bicycle", "bus", "car", "motorcycle", "pedestrian", "trailer", "truck"]

# 99.9 percentile of the l2 velocity error per class / 0.5 s (pub_tracker.py:21-31)
NUSCENE_CLS_VELOCITY_ERROR = {"car": 2, "truck": 2, "bus": 4, "trailer": 2, "pedestrian": 0.75, "motorcycle": 2, "bicycle": 1.5}


def center_greedy_device(problems, device=None, want_dist=True):
    """problems: list of (dets (N,2) f32, tracks (M,2) f32, det_cat (N,) i32, trk_cat (M,) i32, max_diff (N,) f32) numpy
    tuples with N, M >= 1.  Returns a list of (dist (N,M) float64 numpy or None, matched_indices (K,2) int32 numpy,
    row_any (N,) bool, col_any (M,) bool); the two flag vectors say whether a detection / a track has any partner inside
    its gate - all the tracker needs from the matrix unless it runs the Hungarian solver, so with want_dist=False the
    float64 matrices never leave the device."""
    lib = hip.load()
    device = device or torch.device("cuda", torch.cuda.current_device())
    S = len(problems)
    if S == 0:
        return []
    Nmax = max(p[0].shape[0] for p in problems)
    Mmax = max(p[1].shape[0] for p in problems)
    # ONE host buffer -> one copy to the device, one result buffer -> one copy back (4-byte words; seven small copies in and three out
    # cost more than the kernel at the sizes of a frame)
    sizes = [S * Nmax * 2, S * Mmax * 2, S * Nmax, S * Nmax, S * Mmax, S, S]  # det_xy, trk_xy, max_diff | det_cat, trk_cat, n, m
    offs = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    host = torch.zeros(offs[-1], dtype=torch.int32, pin_memory=device.type == "cuda")
    hi, hf = host.numpy(), host.numpy().view(np.float32)
    dxy, txy = hf[offs[0]:offs[1]].reshape(S, Nmax, 2), hf[offs[1]:offs[2]].reshape(S, Mmax, 2)
    md = hf[offs[2]:offs[3]].reshape(S, Nmax)
    dc, tc = hi[offs[3]:offs[4]].reshape(S, Nmax), hi[offs[4]:offs[5]].reshape(S, Mmax)
    n, m = hi[offs[5]:offs[6]], hi[offs[6]:offs[7]]
    for s, (d, t, a, b, g) in enumerate(problems):
        n[s], m[s] = d.shape[0], t.shape[0]
        dxy[s, :n[s]], txy[s, :m[s]], dc[s, :n[s]], tc[s, :m[s]], md[s, :n[s]] = d, t, a, b, g
    dbuf = host.to(device, non_blocking=True)
    dev = [dbuf[offs[i]:offs[i + 1]] for i in range(7)]
    dev = [dev[0].view(torch.float32), dev[1].view(torch.float32), dev[3], dev[4], dev[2].view(torch.float32), dev[5], dev[6]]
    dist = torch.empty(S, Nmax, Mmax, dtype=torch.float64, device=device) if want_dist else None
    out = torch.zeros(S * (2 * Nmax + Mmax), dtype=torch.int32, device=device)  # match | row_any | col_any (the flags start at zero)
    match, row_any, col_any = out[:S * Nmax], out[S * Nmax:2 * S * Nmax], out[2 * S * Nmax:]
    hip.check(lib.shasta_center_greedy_f32(*[hip.ptr(x) for x in dev], S, Nmax, Mmax, hip.ptr(dist), hip.ptr(match), hip.ptr(row_any),
                                           hip.ptr(col_any), hip.stream_ptr()), "shasta_center_greedy_f32")
    out_h = out.cpu().numpy()
    match_h = out_h[:S * Nmax].reshape(S, Nmax)
    row_h = out_h[S * Nmax:2 * S * Nmax].reshape(S, Nmax) != 0
    col_h = out_h[2 * S * Nmax:].reshape(S, Mmax) != 0
    dist_h = dist.cpu().numpy() if want_dist else None
    res = []
    for s in range(S):
        mi = match_h[s, :n[s]]
        rows = np.nonzero(mi >= 0)[0]
        pairs = np.stack([rows, mi[rows]], axis=1).astype(np.int32).reshape(-1, 2)
        res.append((dist_h[s, :n[s], :m[s]].copy() if want_dist else None, pairs, row_h[s, :n[s]].copy(), col_h[s, :m[s]].copy()))
    return res


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

    # ---- the phases of one step (reference: pub_tracker.py:55-210) -----------------------------------------------------
    def _prepare(self, frame_dets, time_lag):
        """Class filter + per-detection annotations (ct, tracking offset, class id) and the arrays of the distance step.
        None for a frame without any detection: the caller then drops every track, as the reference does."""
        if not frame_dets:
            return None
        kept = [d for d in frame_dets if d["detection_name"] in NUSCENES_TRACKING_NAMES]
        for d in kept:
            d["ct"] = np.array(d["translation"][:2])
            d["tracking"] = np.array(d["velocity"][:2]) * -1 * time_lag
            d["label_preds"] = NUSCENES_TRACKING_NAMES.index(d["detection_name"])
        _ = kept[0]  # a frame with detections but none of a tracking class raises IndexError, like the reference
        # predicted previous-frame centres: float32 offset added to the float64 centre, then the whole row to float32
        det_xy = np.array([d["ct"] + d["tracking"].astype(np.float32) for d in kept], np.float32)
        det_cls = np.array([d["label_preds"] for d in kept], np.int32)
        gate = np.array([self.NUSCENE_CLS_VELOCITY_ERROR[d["detection_name"]] for d in kept], np.float32)
        trk_xy = np.array([t["ct"] for t in self.tracks], np.float32)
        trk_cls = np.array([t["label_preds"] for t in self.tracks], np.int32)
        return kept, det_xy, trk_xy, det_cls, trk_cls, gate

    def _finish(self, dets, det_xy, trk_xy, dist, pairs, det_near=None, trk_near=None):
        """Bookkeeping after the assignment: ids, ages, `active` counters, confidence refinement, the newborn / dead
        suppression rules, carrying unmatched tracks for up to max_age frames."""
        n_det, n_trk = det_xy.shape[0], trk_xy.shape[0]
        taken_d, taken_t = set(pairs[:, 0].tolist()), set(pairs[:, 1].tolist())
        free_dets = [i for i in range(n_det) if i not in taken_d]
        free_tracks = [j for j in range(n_trk) if j not in taken_t]
        if det_near is None and dist is not None:  # flags of the newborn / dead rules straight from the matrix
            det_near = np.array([(dist[i, :] <= self.NUSCENE_CLS_VELOCITY_ERROR[dets[i]["detection_name"]]).sum() > 0
                                 for i in range(n_det)], bool)
            trk_near = np.array([(dist[:, j] <= self.NUSCENE_CLS_VELOCITY_ERROR[self.tracks[j]["detection_name"]]).sum() > 0
                                 for j in range(n_trk)], bool)
        if self.hungarian:  # the solver pairs everything: pairs at the invalid cost are not matches
            good = [p for p in pairs if not dist[p[0], p[1]] > 1e16]
            free_dets += [p[0] for p in pairs if dist[p[0], p[1]] > 1e16]
            pairs = np.array(good).reshape(-1, 2)
        out = []
        for i, j in pairs:
            det, old = dets[i], self.tracks[j]
            det["tracking_id"] = old["tracking_id"]
            if self.refine_confidence:
                det["ref_detection_score"] = ((det["ref_detection_score"] > self.alpha) * self.beta * det["detection_score"]
                                              + (1 - self.beta) * old["ref_detection_score"])
            det["age"] = 1
            det["active"] = old["active"] + 1
            out.append(det)
        for i in free_dets:
            det = dets[i]
            # an unmatched detection that is not marked newborn but sits inside the gate of some track is dropped
            if n_trk > 0 and "newborn" not in det and det_near[i]:
                continue
            self.id_count += 1
            det.update(tracking_id=self.id_count, ref_detection_score=det["detection_score"], age=1, active=1)
            out.append(det)
        for j in free_tracks:
            old = self.tracks[j]
            if "dead" in old and trk_near[j]:
                continue
            if old["age"] < self.max_age:  # coast: keep the track, move its centre forward by its last offset
                old["age"] += 1
                old["active"] = 0
                if "tracking" in old:
                    old["ct"] = old["ct"] + old["tracking"] * -1
                out.append(old)
        self.tracks = out
        return out

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
    need_dist = any(trackers[k].hungarian for k in where)  # only the Hungarian solver needs the matrices on the host
    solved = dict(zip(where, center_greedy_device(problems, want_dist=need_dist))) if problems else {}
    outs = []
    for k, trk in enumerate(trackers):
        p = prepared[k]
        if p is None:
            trk.tracks = []
            outs.append([])
            continue
        results, dets, tracks = p[0], p[1], p[2]
        det_near = trk_near = None
        if k in solved:
            dist, matched, det_near, trk_near = solved[k]
            if trk.hungarian:
                dist, matched = trk._host_assign(dist)
        else:
            assert len(trk.tracks) == 0
            dist, matched = None, np.array([], np.int32).reshape(-1, 2)
        outs.append(trk._finish(results, dets, tracks, dist, matched, det_near, trk_near))
    return outs


# per-class confidence-refinement parameters of the merged tracker (pub_tracker_merged.py:34-42)
TRK_REF = {
    "bicycle": {"alpha": 0.5, "beta": 0.4, "ref": True},
    "bus": {"alpha": 0.5, "beta": 0.7, "ref": True},
    "car": {"alpha": 0.5, "beta": 0.5, "ref": True},
    "motorcycle": {"alpha": 0.5, "beta": 0.5, "ref": True},
    "pedestrian": {"alpha": 0.5, "beta": 0.5, "ref": True},
    "trailer": {"alpha": 0.5, "beta": 0.4, "ref": True},
    "truck": {"alpha": 0.5, "beta": 0.5, "ref": True},
}


class PubTrackerMerged(object):
    """tools/nusc_shasta/pub_tracker_merged.py:57-225, the tracker `pub_test.py` runs over the merged seven-class detections
    (official_val.sh): one association per tracking class and frame, per-class confidence refinement (TRK_REF), unmatched
    tracks coast with a decayed score.  Same `reset()` / `step_centertrack(results, time_lag)` contract; the (up to seven)
    class problems of a frame - of many scenes with `step_batch_merged` - are ONE launch of the centre-distance / greedy kernel."""

    def __init__(self, hungarian=False, max_age=0, verbose=False):
        self.hungarian = hungarian
        self.max_age = max_age
        self.NUSCENE_CLS_VELOCITY_ERROR = NUSCENE_CLS_VELOCITY_ERROR
        self.trk_ref = TRK_REF
        if verbose:
            print("Use hungarian: {}".format(hungarian))
            print(self.NUSCENE_CLS_VELOCITY_ERROR)
        self.reset()

    def reset(self):
        self.id_count = 0
        self.tracks = []

    def _prepare(self, results, time_lag):
        """Per class: (class name, detections, tracks of that class, arrays of the distance step or None).  One pass over the frame's
        detections groups them by class and one array operation per class forms every detection's `ct` / `tracking` (the values of
        pub_tracker_merged.py:92-99, element for element; the per-detection arrays are rows of the class arrays)."""
        by_class = {}
        for i, d in enumerate(results):
            by_class.setdefault(d["detection_name"], []).append(i)
        all_ct = np.array([d["translation"][:2] for d in results])                      # one array per frame, sliced per class below
        all_trk = np.array([d["velocity"][:2] for d in results]) * -1 * time_lag
        tracks_by_class = {}
        for t in self.tracks:
            tracks_by_class.setdefault(t["detection_name"], []).append(t)
        per_class = []
        for label, name in enumerate(NUSCENES_TRACKING_NAMES):
            idx = by_class.get(name)
            if not idx:  # pub_tracker_merged.py:101-102: nothing of this class in the frame -> its tracks are dropped
                continue
            dets = [results[i] for i in idx]
            ct, trk = all_ct[idx], all_trk[idx]
            for i, d in enumerate(dets):
                d["ct"], d["tracking"], d["label_preds"] = ct[i], trk[i], label
            tracks = tracks_by_class.get(name, [])
            det_xy = (ct + trk.astype(np.float32)).astype(np.float32)
            det_cls = np.full(len(dets), label, np.int32)
            gate = np.full(len(dets), self.NUSCENE_CLS_VELOCITY_ERROR[name], np.float32)
            trk_xy = np.array([t["ct"] for t in tracks], np.float32)
            trk_cls = np.array([t["label_preds"] for t in tracks], np.int32)
            per_class.append((name, dets, tracks, det_xy, trk_xy, det_cls, trk_cls, gate))
        return per_class

    def _finish_class(self, name, dets, tracks, dist, pairs, det_near, trk_near, ret):
        n_det, n_trk = len(dets), len(tracks)
        taken_d, taken_t = set(pairs[:, 0].tolist()), set(pairs[:, 1].tolist())
        free_dets = [i for i in range(n_det) if i not in taken_d]
        free_tracks = [j for j in range(n_trk) if j not in taken_t]
        if det_near is None and dist is not None:
            gate = self.NUSCENE_CLS_VELOCITY_ERROR[name]
            det_near = np.array([(dist[i, :] <= gate).sum() > 0 for i in range(n_det)], bool)
            trk_near = np.array([(dist[:, j] <= gate).sum() > 0 for j in range(n_trk)], bool)
        if self.hungarian:
            good = [p for p in pairs if not dist[p[0], p[1]] > 1e16]
            free_dets += [p[0] for p in pairs if dist[p[0], p[1]] > 1e16]
            pairs = np.array(good).reshape(-1, 2)
        ref = self.trk_ref[name]
        for i, j in pairs:
            det, old = dets[i], tracks[j]
            det["tracking_id"] = old["tracking_id"]
            if ref["ref"]:
                det["ref_detection_score"] = ((det["ref_detection_score"] > ref["alpha"]) * ref["beta"] * det["detection_score"]
                                              + (1 - ref["beta"]) * old["ref_detection_score"])
            else:
                det["ref_detection_score"] = det["detection_score"]
            det["age"] = 1
            det["active"] = old["active"] + 1
            ret.append(det)
        for i in free_dets:
            det = dets[i]
            if n_trk > 0 and "newborn" not in det and det_near[i]:
                continue
            self.id_count += 1
            det["tracking_id"] = self.id_count
            det["ref_detection_score"] = ref["beta"] * det["detection_score"] if ref["ref"] else det["detection_score"]
            det["age"] = 1
            det["active"] = 1
            ret.append(det)
        for j in free_tracks:
            old = tracks[j]
            if "dead" in old and trk_near[j]:
                continue
            if old["age"] < self.max_age:
                old["age"] += 1
                old["active"] = 0
                if ref["ref"]:
                    old["ref_detection_score"] = (1 - ref["beta"]) * old["ref_detection_score"]
                if "tracking" in old:
                    old["ct"] = old["ct"] + old["tracking"] * -1
                ret.append(old)

    def _host_assign(self, dist):
        return PubTracker._host_assign(self, dist)

    def step_centertrack(self, results, time_lag):
        return step_batch_merged([self], [results], [time_lag])[0]


def step_batch_merged(trackers, results_list, time_lags):
    """One merged-tracker step for many independent scenes: every (scene, class) association is a problem of ONE launch."""
    prepared, problems, where = [], [], []
    for k, (trk, results, lag) in enumerate(zip(trackers, results_list, time_lags)):
        if len(results) == 0:
            prepared.append(None)
            continue
        per_class = trk._prepare(results, lag)
        prepared.append(per_class)
        for c, p in enumerate(per_class):
            if len(p[4]) > 0:
                where.append((k, c))
                problems.append((p[3], p[4], p[5], p[6], p[7]))
    need_dist = any(trackers[k].hungarian for k, _ in where)
    solved = dict(zip(where, center_greedy_device(problems, want_dist=need_dist))) if problems else {}
    outs = []
    for k, trk in enumerate(trackers):
        per_class = prepared[k]
        if per_class is None:
            trk.tracks = []
            outs.append([])
            continue
        ret = []
        for c, (name, dets, tracks, det_xy, trk_xy, _, _, _) in enumerate(per_class):
            det_near = trk_near = None
            if (k, c) in solved:
                dist, matched, det_near, trk_near = solved[(k, c)]
                if trk.hungarian:
                    dist, matched = trk._host_assign(dist)
            else:
                assert len(tracks) == 0
                dist, matched = None, np.array([], np.int32).reshape(-1, 2)
            trk._finish_class(name, dets, tracks, dist, matched, det_near, trk_near, ret)
        trk.tracks = ret
        outs.append(ret)
    return outs


def track_scenes_merged_device(scene_frames, max_age=0, device=None, plain=False, refine_confidence=False, alpha=0.5, beta=0.5):
    """track_scenes_collect(track_scenes_launch(...)): see there."""
    return track_scenes_collect(track_scenes_launch(scene_frames, max_age=max_age, device=device, plain=plain, refine_confidence=refine_confidence,
                                                    alpha=alpha, beta=beta))


def track_scenes_launch(scene_frames, max_age=0, device=None, plain=False, refine_confidence=False, alpha=0.5, beta=0.5, stream=None):
    """The merged tracker (PubTrackerMerged, greedy; plain=True: PubTracker with its refine_confidence / alpha / beta - one list over all
    tracking classes, result order matched detections then new ones) for whole scenes in ONE launch (csrc/track.hip `track_merged_kernel`,
    shasta_track_merged_f64): scene_frames = [[(detections of the frame: list of nuScenes-format dicts with `ref_detection_score`,
    time_lag), ...] per scene].  Returns per scene, per frame, the result rows' sources in the order pub_test.py emits them:
    a list of (detection dict, tracking_id, refined ref_detection_score) - class by class, matched detections then new ones - or None
    when a scene exceeds the kernel's capacities (512 detections per frame, 768 tracks alive): the caller then takes the per-frame path.
    The dicts are NOT modified (the host tracker annotates them in place).
    This half packs the inputs, queues one copy in, the kernel and one copy out on `stream` (default: the current one) and returns a handle
    at once; track_scenes_collect(handle) waits for the copy out and builds the rows - the chain launches a scene's tracker on a side
    stream while the device works on the next scenes' maps."""
    import ctypes as C
    lib = hip.load()
    device = device or torch.device("cuda", torch.cuda.current_device())
    S = len(scene_frames)
    if S == 0:
        return dict(done=[])
    label = {n: i for i, n in enumerate(NUSCENES_TRACKING_NAMES)}
    Fmax = max(1, max(len(fr) for fr in scene_frames))
    dets = [d for fr in scene_frames for (ds, _) in fr for d in ds]
    D = len(dets)
    off = np.zeros((S, Fmax + 1), np.int32)
    lag = np.zeros((S, Fmax), np.float64)
    nfr = np.zeros(S, np.int32)
    g = 0
    for s, fr in enumerate(scene_frames):
        nfr[s] = len(fr)
        for f, (ds, tl) in enumerate(fr):
            off[s, f] = g
            g += len(ds)
            lag[s, f] = tl
        off[s, len(fr):] = g
    if D == 0:
        return dict(done=[[[] for _ in fr] for fr in scene_frames])
    xy = np.array([d["translation"][:2] for d in dets], np.float64).reshape(D, 2)
    vel = np.array([d["velocity"][:2] for d in dets], np.float64).reshape(D, 2)
    cls = np.array([label.get(d["detection_name"], -1) for d in dets], np.int32)
    score = np.array([d["detection_score"] for d in dets], np.float64)
    ref = np.array([d.get("ref_detection_score", 0.0) for d in dets], np.float64) if plain else np.array([d["ref_detection_score"] for d in dets], np.float64)
    flags = np.array([("newborn" in d) | (("dead" in d) << 1) for d in dets], np.int32)
    # one 8-byte-word buffer in, one back
    words = [2 * D, 2 * D, D, D, (D + 1) // 2, (D + 1) // 2, (off.size + 1) // 2, lag.size, (S + 1) // 2]
    o = np.concatenate([[0], np.cumsum(words)]).tolist()
    host = torch.zeros(o[-1], dtype=torch.float64, pin_memory=device.type == "cuda")
    h = host.numpy()
    h[o[0]:o[1]] = xy.ravel()
    h[o[1]:o[2]] = vel.ravel()
    h[o[2]:o[3]] = score
    h[o[3]:o[4]] = ref
    h[o[4]:o[5]].view(np.int32)[:D] = cls
    h[o[5]:o[6]].view(np.int32)[:D] = flags
    h[o[6]:o[7]].view(np.int32)[:off.size] = off.ravel()
    h[o[7]:o[8]] = lag.ravel()
    h[o[8]:o[9]].view(np.int32)[:S] = nfr
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    with ctx:
        dbuf = host.to(device, non_blocking=True)
        seg = [dbuf[o[i]:o[i + 1]] for i in range(9)]
        out = torch.zeros(D + (D + 1) // 2 * 2 + (S + 1) // 2, dtype=torch.float64, device=device)  # ref | status | id | err
        o_ref, o_st = out[:D], out[D:D + (D + 1) // 2].view(torch.int32)
        o_id, o_err = out[D + (D + 1) // 2:D + 2 * ((D + 1) // 2)].view(torch.int32), out[D + 2 * ((D + 1) // 2):].view(torch.int32)
        names = NUSCENES_TRACKING_NAMES
        gate = (C.c_float * len(names))(*[float(NUSCENE_CLS_VELOCITY_ERROR[n]) for n in names])
        if plain:
            refon = (C.c_int32 * len(names))(*([int(bool(refine_confidence))] * len(names)))
            alpha = (C.c_double * len(names))(*([float(alpha)] * len(names)))
            beta = (C.c_double * len(names))(*([float(beta)] * len(names)))
        else:
            refon = (C.c_int32 * len(names))(*[int(bool(TRK_REF[n]["ref"])) for n in names])
            alpha = (C.c_double * len(names))(*[float(TRK_REF[n]["alpha"]) for n in names])
            beta = (C.c_double * len(names))(*[float(TRK_REF[n]["beta"]) for n in names])
        rc = lib.shasta_track_merged_f64(hip.ptr(seg[0]), hip.ptr(seg[1]), hip.ptr(seg[4].view(torch.int32)), hip.ptr(seg[2]), hip.ptr(seg[3]),
                                         hip.ptr(seg[5].view(torch.int32)), hip.ptr(seg[6].view(torch.int32)), hip.ptr(seg[7]),
                                         hip.ptr(seg[8].view(torch.int32)), S, Fmax, len(names), gate, refon, alpha, beta, int(max_age), int(bool(plain)),
                                         hip.ptr(o_st), hip.ptr(o_id), hip.ptr(o_ref), hip.ptr(o_err), hip.stream_ptr())
        if rc == hip.E_UNSUPPORTED:  # a device that does not grant the kernel its LDS: every scene takes the per-frame path
            return dict(done=[None] * S)
        hip.check(rc, "shasta_track_merged_f64")
        back = torch.empty(out.shape, dtype=out.dtype, pin_memory=device.type == "cuda")
        back.copy_(out, non_blocking=True)
        ev = None
        if device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
    return dict(back=back, ev=ev, keep=(host, dbuf, out), D=D, S=S, dets=dets, off=off, nfr=nfr, cls=cls, plain=plain)


def track_scenes_collect(handle):
    """Second half of track_scenes_merged_device: per scene, per frame, [(detection dict, tracking id, refined score)] or None (see
    track_scenes_launch)."""
    if "done" in handle:
        return handle["done"]
    if handle["ev"] is not None:
        handle["ev"].synchronize()
    D, S, dets, off, nfr, cls, plain = (handle[k] for k in ("D", "S", "dets", "off", "nfr", "cls", "plain"))
    oh = handle["back"].numpy()
    r_ref = oh[:D].tolist()
    r_st = oh[D:D + (D + 1) // 2].view(np.int32)[:D]
    r_id = oh[D + (D + 1) // 2:D + 2 * ((D + 1) // 2)].view(np.int32)[:D].tolist()
    r_err = oh[D + 2 * ((D + 1) // 2):].view(np.int32)[:S]
    # result order inside a frame: class, then matched before new, then file order - one sort for the whole split
    frame_of = np.zeros(D, np.int64)
    bounds, fid = [], 0
    for s in range(S):
        for f in range(int(nfr[s])):
            frame_of[off[s, f]:off[s, f + 1]] = fid
            bounds.append((s, f))
            fid += 1
    kept = np.nonzero(r_st > 0)[0]
    group = np.zeros(D, np.int32) if plain else cls  # the plain tracker's result is one list: matched, then new
    kept = kept[np.lexsort((kept, r_st[kept], group[kept], frame_of[kept]))]
    per_frame = np.bincount(frame_of[kept], minlength=fid) if fid else np.zeros(0, np.int64)
    kept = kept.tolist()
    res = [None if r_err[s] != 0 else [None] * int(nfr[s]) for s in range(S)]
    p0 = 0
    for k, (s, f) in enumerate(bounds):
        cnt = int(per_frame[k])
        if res[s] is not None:
            res[s][f] = [(dets[i], r_id[i], r_ref[i]) for i in kept[p0:p0 + cnt]]
        p0 += cnt
    return res
