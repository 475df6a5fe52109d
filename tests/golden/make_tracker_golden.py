"""Generates tests/golden/pub_tracker_golden.json.gz: synthetic detection sequences and what the REFERENCE's PubTracker
(tools/nusc_shasta/pub_tracker.py, imported in place; build container only) returns for them frame by frame.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_tracker_golden.py
"""
import contextlib
import copy
import io
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SHASTA_REFERENCE", "/root/reference")
NAMES = ["bicycle", "bus", "car", "motorcycle", "pedestrian", "trailer", "truck", "barrier", "traffic_cone"]

CASES = [dict(hungarian=False, max_age=0, refine_confidence=False), dict(hungarian=False, max_age=3, refine_confidence=True),
         dict(hungarian=True, max_age=2, refine_confidence=False), dict(hungarian=False, max_age=1, refine_confidence=True, alpha=0.3, beta=0.7)]


def synth_scene(rng, n_frames, n_obj):
    """Objects move with constant velocity + noise; detections drop out, false positives appear, some carry the
    'newborn' / 'dead' marks the affinity decode adds (tools/nusc_shasta/eval.py:127-173)."""
    pos = rng.uniform(-40, 40, (n_obj, 2))
    vel = rng.normal(0, 2.0, (n_obj, 2))
    cls = rng.integers(0, len(NAMES), n_obj)
    frames, uid = [], 0
    for f in range(n_frames):
        dets = []
        if f == 3:  # an empty frame: the tracker forgets everything
            frames.append(dets)
            pos = pos + 0.5 * vel
            continue
        for o in range(n_obj):
            if rng.uniform() < 0.15:
                continue
            p = pos[o] + rng.normal(0, 0.15, 2)
            d = dict(uid=uid, detection_name=NAMES[int(cls[o])], translation=[float(p[0]), float(p[1]), 0.5],
                     velocity=[float(vel[o, 0] + rng.normal(0, 0.2)), float(vel[o, 1] + rng.normal(0, 0.2))],
                     detection_score=float(rng.uniform(0.05, 1)), ref_detection_score=float(rng.uniform(0, 1)))
            if rng.uniform() < 0.1:
                d["newborn"] = True
            if rng.uniform() < 0.1:
                d["dead"] = True
            uid += 1
            dets.append(d)
        for _ in range(int(rng.integers(0, 4))):  # clutter
            dets.append(dict(uid=uid, detection_name=NAMES[int(rng.integers(0, 7))],
                             translation=[float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40)), 0.5],
                             velocity=[float(rng.normal()), float(rng.normal())], detection_score=float(rng.uniform(0.05, 0.5)),
                             ref_detection_score=float(rng.uniform(0, 1))))
            uid += 1
        order = rng.permutation(len(dets))
        frames.append([dets[i] for i in order])
        pos = pos + 0.5 * vel
    return frames


def snapshot(ret):
    return [dict(uid=t["uid"], tracking_id=int(t["tracking_id"]), age=int(t["age"]), active=int(t["active"]),
                 ref_detection_score=float(t["ref_detection_score"]), ct=[float(t["ct"][0]), float(t["ct"][1])]) for t in ret]


def main():
    sys.path.insert(0, os.path.join(REF, "tools", "nusc_shasta"))
    from pub_tracker import PubTracker  # the reference class
    rng = np.random.default_rng(7)
    scenes = [synth_scene(rng, 8, n) for n in (5, 30, 120)]
    out = dict(scenes=scenes, cases=CASES, expected=[])
    for case in CASES:
        per_case = []
        for frames in scenes:
            with contextlib.redirect_stdout(io.StringIO()):
                trk = PubTracker(**case)
            per_scene = []
            for dets in copy.deepcopy(frames):
                per_scene.append(snapshot(trk.step_centertrack(dets, 0.5)))
            per_case.append(per_scene)
        out["expected"].append(per_case)
    import gzip
    path = os.path.join(HERE, "pub_tracker_golden.json.gz")
    if "merged" not in sys.argv[1:]:
        with gzip.open(path, "wt", compresslevel=9) as f:
            json.dump(out, f)
        print("wrote", path, os.path.getsize(path), "bytes")
    # the merged (all-class) tracker of pub_test.py: tools/nusc_shasta/pub_tracker_merged.py, same scenes
    from pub_tracker_merged import PubTrackerMerged
    merged_cases = [dict(hungarian=False, max_age=0), dict(hungarian=False, max_age=3), dict(hungarian=True, max_age=2)]
    mout = dict(cases=merged_cases, expected=[])
    for case in merged_cases:
        per_case = []
        for frames in scenes:
            with contextlib.redirect_stdout(io.StringIO()):
                trk = PubTrackerMerged(**case)
            per_case.append([snapshot(trk.step_centertrack(dets, 0.5)) for dets in copy.deepcopy(frames)])
        mout["expected"].append(per_case)
    path = os.path.join(HERE, "pub_tracker_merged_golden.json.gz")
    with gzip.open(path, "wt", compresslevel=9) as f:
        json.dump(mout, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
