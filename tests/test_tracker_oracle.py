"""CPU: the tracker oracle (oracle/tracker_oracle.py) against the reference classes' frame-by-frame output on the golden scenes
(PubTracker and PubTrackerMerged, tests/golden/make_tracker_golden.py).  This pins the checker the pipeline tests use."""
import copy
import gzip
import json
import os

import numpy as np
import pytest

from oracle import tracker_oracle as TO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with gzip.open(os.path.join(G, name), "rt") as f:
        return json.load(f)


def _same(ret, want):
    assert len(ret) == len(want)
    for t, w in zip(ret, want):
        assert (t["uid"], int(t["tracking_id"]), int(t["age"]), int(t["active"])) == (w["uid"], w["tracking_id"], w["age"], w["active"])
        assert abs(float(t["ref_detection_score"]) - w["ref_detection_score"]) <= 1e-12
        assert np.allclose([float(t["ct"][0]), float(t["ct"][1])], w["ct"], rtol=0, atol=1e-12)


def test_pub_tracker_oracle_matches_reference():
    g = _load("pub_tracker_golden.json.gz")
    for ci, case in enumerate(g["cases"]):
        for si, frames in enumerate(g["scenes"]):
            trk = TO.PubTrackerOracle(**case)
            for fi, dets in enumerate(copy.deepcopy(frames)):
                _same(trk.step_centertrack(dets, 0.5), g["expected"][ci][si][fi])


def test_merged_tracker_oracle_matches_reference():
    g, m = _load("pub_tracker_golden.json.gz"), _load("pub_tracker_merged_golden.json.gz")
    for ci, case in enumerate(m["cases"]):
        for si, frames in enumerate(g["scenes"]):
            trk = TO.PubTrackerMergedOracle(**case)
            for fi, dets in enumerate(copy.deepcopy(frames)):
                _same(trk.step_centertrack(dets, 0.5), m["expected"][ci][si][fi])
