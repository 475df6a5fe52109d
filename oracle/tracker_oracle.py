"""CPU oracle of the public tracker step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-numpy restatement of tools/nusc_shasta/pub_tracker.py:35-210 (`PubTracker.step_centertrack`),
tools/nusc_shasta/pub_tracker_merged.py:57-225 (`PubTrackerMerged.step_centertrack`) and
tools/nusc_shasta/track_utils.py:3-14 (`greedy_assignment`).  Parity pin: the reference's own classes, run in the build
container on synthetic scenes (tests/golden/make_tracker_golden.py -> pub_tracker_golden.json.gz,
pub_tracker_merged_golden.json.gz), checked by tests/test_tracker_oracle.py.  Only tests/ and the pipeline checker
(oracle/pipeline_oracle.py) import this."""
import copy

import numpy as np

NAMES = ["bicycle", "bus", "car", "motorcycle", "pedestrian", "trailer", "truck"]
VEL_ERR = {"car": 2, "truck": 2, "bus": 4, "trailer": 2, "pedestrian": 0.75, "motorcycle": 2, "bicycle": 1.5}
TRK_REF = {"bicycle": (0.5, 0.4), "bus": (0.5, 0.7), "car": (0.5, 0.5), "motorcycle": (0.5, 0.5), "pedestrian": (0.5, 0.5),
           "trailer": (0.5, 0.4), "truck": (0.5, 0.5)}  # (alpha, beta), refinement on for every class (pub_tracker_merged.py:34-42)


def greedy_assignment(dist):
    """track_utils.py:3-14: rows in order, first minimum of the row, the column is then closed for later rows."""
    pairs = []
    if dist.shape[1] == 0:
        return np.array(pairs, np.int32).reshape(-1, 2)
    for i in range(dist.shape[0]):
        j = dist[i].argmin()
        if dist[i][j] < 1e16:
            dist[:, j] = 1e18
            pairs.append([i, j])
    return np.array(pairs, np.int32).reshape(-1, 2)


def _associate(results, tracks, hungarian):
    """pub_tracker.py:78-128: float32 centres, float64 distance, class / velocity gate, greedy or Hungarian."""
    N, M = len(results), len(tracks)
    dets = np.array([d["ct"] + d["tracking"].astype(np.float32) for d in results], np.float32)
    item_cat = np.array([d["label_preds"] for d in results], np.int32)
    track_cat = np.array([t["label_preds"] for t in tracks], np.int32)
    max_diff = np.array([VEL_ERR[d["detection_name"]] for d in results], np.float32)
    trk = np.array([t["ct"] for t in tracks], np.float32)
    dist = None
    if len(trk) > 0:
        dist = np.sqrt(((trk.reshape(1, -1, 2) - dets.reshape(-1, 1, 2)) ** 2).sum(axis=2))
        invalid = ((dist > max_diff.reshape(N, 1)) + (item_cat.reshape(N, 1) != track_cat.reshape(1, M))) > 0
        dist = dist + invalid * 1e18
        if hungarian:
            from scipy.optimize import linear_sum_assignment
            dist[dist > 1e18] = 1e18
            r, c = linear_sum_assignment(copy.deepcopy(dist))
            matched = np.concatenate((r.reshape(-1, 1), c.reshape(-1, 1)), axis=-1)
        else:
            matched = greedy_assignment(copy.deepcopy(dist))
    else:
        matched = np.array([], np.int32).reshape(-1, 2)
    un_d = [d for d in range(dets.shape[0]) if d not in matched[:, 0]]
    un_t = [t for t in range(trk.shape[0]) if t not in matched[:, 1]]
    if hungarian:
        good = []
        for m in matched:
            if dist[m[0], m[1]] > 1e16:
                un_d.append(m[0])
            else:
                good.append(m)
        matched = np.array(good).reshape(-1, 2)
    return dist, matched, un_d, un_t, len(trk)


def _annotate(det, time_lag):
    det["ct"] = np.array(det["translation"][:2])
    det["tracking"] = np.array(det["velocity"][:2]) * -1 * time_lag
    det["label_preds"] = NAMES.index(det["detection_name"])


class PubTrackerOracle:
    def __init__(self, hungarian=False, max_age=0, refine_confidence=False, alpha=0.5, beta=0.5):
        self.hungarian, self.max_age = hungarian, max_age
        self.refine_confidence, self.alpha, self.beta = refine_confidence, alpha, beta
        self.reset()

    def reset(self):
        self.id_count, self.tracks = 0, []

    def step_centertrack(self, results, time_lag):
        if len(results) == 0:
            self.tracks = []
            return []
        kept = []
        for det in results:
            if det["detection_name"] not in NAMES:
                continue
            _annotate(det, time_lag)
            kept.append(det)
        results = kept
        _ = results[0]
        dist, matched, un_d, un_t, M = _associate(results, self.tracks, self.hungarian)
        ret = []
        for m in matched:
            trk = results[m[0]]
            old = self.tracks[m[1]]
            trk["tracking_id"] = old["tracking_id"]
            if self.refine_confidence:
                trk["ref_detection_score"] = ((trk["ref_detection_score"] > self.alpha) * self.beta * trk["detection_score"]
                                              + (1 - self.beta) * old["ref_detection_score"])
            trk["age"] = 1
            trk["active"] = old["active"] + 1
            ret.append(trk)
        for i in un_d:
            trk = results[i]
            if M > 0 and "newborn" not in trk and (dist[i, :] <= VEL_ERR[trk["detection_name"]]).sum():
                continue
            self.id_count += 1
            trk.update(tracking_id=self.id_count, ref_detection_score=trk["detection_score"], age=1, active=1)
            ret.append(trk)
        for i in un_t:
            trk = self.tracks[i]
            if "dead" in trk and (dist[:, i] <= VEL_ERR[trk["detection_name"]]).sum():
                continue
            if trk["age"] < self.max_age:
                trk["age"] += 1
                trk["active"] = 0
                if "tracking" in trk:
                    trk["ct"] = trk["ct"] + trk["tracking"] * -1
                ret.append(trk)
        self.tracks = ret
        return ret


class PubTrackerMergedOracle:
    def __init__(self, hungarian=False, max_age=0):
        self.hungarian, self.max_age = hungarian, max_age
        self.reset()

    def reset(self):
        self.id_count, self.tracks = 0, []

    def step_centertrack(self, results, time_lag):
        if len(results) == 0:
            self.tracks = []
            return []
        ret = []
        for name in NAMES:
            cur = []
            for det in results:
                if det["detection_name"] != name:
                    continue
                _annotate(det, time_lag)
                cur.append(det)
            tracks = [t for t in self.tracks if t["detection_name"] == name]
            if len(cur) == 0:
                continue
            dist, matched, un_d, un_t, M = _associate(cur, tracks, self.hungarian)
            alpha, beta = TRK_REF[name]
            for m in matched:
                trk, old = cur[m[0]], tracks[m[1]]
                trk["tracking_id"] = old["tracking_id"]
                trk["ref_detection_score"] = ((trk["ref_detection_score"] > alpha) * beta * trk["detection_score"]
                                              + (1 - beta) * old["ref_detection_score"])
                trk["age"] = 1
                trk["active"] = old["active"] + 1
                ret.append(trk)
            for i in un_d:
                trk = cur[i]
                if M > 0 and "newborn" not in trk and (dist[i, :] <= VEL_ERR[name]).sum():
                    continue
                self.id_count += 1
                trk.update(tracking_id=self.id_count, ref_detection_score=beta * trk["detection_score"], age=1, active=1)
                ret.append(trk)
            for i in un_t:
                trk = tracks[i]
                if "dead" in trk and (dist[:, i] <= VEL_ERR[name]).sum():
                    continue
                if trk["age"] < self.max_age:
                    trk["age"] += 1
                    trk["active"] = 0
                    trk["ref_detection_score"] = (1 - beta) * trk["ref_detection_score"]
                    if "tracking" in trk:
                        trk["ct"] = trk["ct"] + trk["tracking"] * -1
                    ret.append(trk)
        self.tracks = ret
        return ret
