"""GPU-box helper: cProfile of the configs 2-4 chain (host side) on the 20 x 40 synthetic split.
    python tools/profile_pipeline_host.py [--top 40]"""
import argparse
import atexit
import shutil
import cProfile
import os
import pstats
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shasta_amd import pipeline, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    root = tempfile.mkdtemp(prefix="shasta_split_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    atexit.register(shutil.rmtree, root, ignore_errors=True)  # the split lives in RAM (tmpfs): never leave it behind
    paths, sc = scenes.write_synthetic_split(root, n_scenes=20, frames_per_scene=40, seed=3)
    models = {n: pipeline.build_class_model(n, dev, seed=1) for n in pipeline.CLASS_CONFIGS}
    neck = scenes.TokenNeck()
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipeline.run_split(models, paths, sc, neck, dev, batch_pairs=40)
        torch.cuda.synchronize()
        print("un-profiled: %.3f s" % (time.perf_counter() - t0), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    pipeline.run_split(models, paths, sc, neck, dev, batch_pairs=40)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(a.top)
    st.sort_stats("cumulative").print_stats(a.top)


if __name__ == "__main__":
    main()
