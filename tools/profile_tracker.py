import atexit
import shutil
"""GPU-box helper: where the merged tracker's time goes on the 20 x 40 synthetic split (after one chain run has produced `merged`).
    python tools/profile_tracker.py"""
import cProfile
import gc
import os
import pstats
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shasta_amd import pipeline, pub_tracker, scenes  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    root = tempfile.mkdtemp(prefix="shasta_split_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    atexit.register(shutil.rmtree, root, ignore_errors=True)  # the split lives in RAM (tmpfs): never leave it behind
    paths, sc = scenes.write_synthetic_split(root, n_scenes=20, frames_per_scene=40, seed=3)
    models = {n: pipeline.build_class_model(n, dev, seed=1) for n in pipeline.CLASS_CONFIGS}
    _, merged, _ = pipeline.run_split(models, paths, sc, scenes.TokenNeck(), dev, batch_pairs=40)
    import json
    meta = json.load(open(paths["frames_meta_path"]))["frames"]
    gc.disable()
    acc = {"prepare": 0.0, "device": 0.0, "finish": 0.0}
    orig_prepare, orig_dev, orig_fin = pub_tracker.PubTrackerMerged._prepare, pub_tracker.center_greedy_device, pub_tracker.PubTrackerMerged._finish_class

    def timed(name, fn):
        def w(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[name] += time.perf_counter() - t0
        return w
    pub_tracker.PubTrackerMerged._prepare = timed("prepare", orig_prepare)
    pub_tracker.center_greedy_device = timed("device", orig_dev)
    pub_tracker.PubTrackerMerged._finish_class = timed("finish", orig_fin)
    for rep in range(3):
        for k in acc:
            acc[k] = 0.0
        preds = {tok: [dict(d) for d in annos] for tok, annos in merged["results"].items()}
        t0 = time.perf_counter()
        pipeline.run_tracking(preds, meta, max_age=4)
        total = time.perf_counter() - t0
        print("tracker %.3f s: " % total + ", ".join("%s %.3f" % kv for kv in acc.items()) + ", rest %.3f" % (total - sum(acc.values())), flush=True)


if __name__ == "__main__":
    main()
