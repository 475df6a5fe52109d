"""Generates tests/golden/sampler_golden.json: index lists of the REFERENCE's DistributedGroupSampler
(det3d/datasets/loader/sampler.py:139-223, loaded in place from /root/reference; build container only) for a few dataset
shapes, epochs and world sizes.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_sampler_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SHASTA_REFERENCE", "/root/reference")


def load_reference_sampler():
    # sampler.py only needs get_dist_info from its package: provide that one name instead of importing all of det3d
    for name in ("det3d", "det3d.torchie", "det3d.torchie.trainer"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["det3d.torchie.trainer"].get_dist_info = lambda: (0, 1)
    spec = importlib.util.spec_from_file_location("_ref_sampler", os.path.join(REF, "det3d/datasets/loader/sampler.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _DS:
    def __init__(self, flag):
        self.flag = np.asarray(flag, np.uint8)

    def __len__(self):
        return len(self.flag)


def main():
    S = load_reference_sampler()
    rng = np.random.default_rng(0)
    cases = []
    for n, groups, spg, world in [(37, 1, 4, 2), (100, 2, 4, 8), (9, 1, 4, 8), (64, 1, 2, 4), (23, 3, 3, 2)]:
        flag = (rng.integers(0, groups, n) if groups > 1 else np.zeros(n, np.int64)).tolist()
        for epoch in (0, 3):
            per_rank, raises = [], False
            for rank in range(world):
                s = S.DistributedGroupSampler(_DS(flag), samples_per_gpu=spg, num_replicas=world, rank=rank)
                s.set_epoch(epoch)
                try:
                    idx = [int(i) for i in s]
                except AssertionError:  # a group smaller than half its padded size cannot be padded by one self-concatenation
                    raises = True
                    break
                assert len(idx) == len(s)
                per_rank.append(idx)
            cases.append(dict(flag=flag, samples_per_gpu=spg, world=world, epoch=epoch, indices=per_rank, raises=raises))
    with open(os.path.join(HERE, "sampler_golden.json"), "w") as f:
        json.dump(cases, f)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
