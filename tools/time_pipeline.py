"""GPU-box helper: configs 2-4 chain on a synthetic split, wall time per stage.
    python tools/time_pipeline.py [--scenes 20] [--frames 40] [--batch 32] [0|1]"""
import argparse
import atexit
import shutil
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shasta_amd import pipeline, scenes  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=20)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--maps", choices=["neck", "features"], default="neck", help="neck: (512,180,180) maps on the device + K0 for all heads; features: TokenBev (CPU stand-in)")
    ap.add_argument("--sync", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    root = tempfile.mkdtemp(prefix="shasta_split_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    atexit.register(shutil.rmtree, root, ignore_errors=True)  # the split lives in RAM (tmpfs): never leave it behind
    t0 = time.perf_counter()
    paths, sc = scenes.write_synthetic_split(root, n_scenes=a.scenes, frames_per_scene=a.frames, seed=3)
    print("split written in %.2f s: %d frames" % (time.perf_counter() - t0, a.scenes * a.frames), flush=True)
    try:
        with open("/proc/cpuinfo") as fh:
            print("host cpu:", next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "?"), flush=True)
    except OSError:
        pass
    models = {n: pipeline.build_class_model(n, dev, seed=1) for n in pipeline.CLASS_CONFIGS}
    bev = scenes.TokenNeck() if a.maps == "neck" else scenes.TokenBev()
    kw = {}
    for rep in range(3):
        timer = pipeline.StageTimer(sync=bool(a.sync))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pipeline.run_split(models, paths, sc, bev, dev, batch_pairs=a.batch, timer=timer, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = a.scenes * a.frames
        print(json.dumps({"rep": rep, "frames": n, "seconds": dt, "frames_per_s": n / dt, "class_frame_pairs_per_s": 7 * n / dt,
                          "stages": {k: round(v, 4) for k, v in timer.seconds.items()}}), flush=True)


if __name__ == "__main__":  # (loader processes are spawned: they import this module again)
    main()
