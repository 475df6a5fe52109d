"""Generates tests/golden/frames_golden.npz: synthetic per-frame detection files in the reference's on-disk formats and what
the REFERENCE's `NuScenesDataset.get_sensor_data` (det3d/datasets/nuscenes/nuscenes.py:198-349) builds from them, for the
input-format loader shasta_amd/frames.py.  Build container only (imports /root/reference in place, nothing copied):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_frames_golden.py

Environment shims, none of them a change to the reference: absent third-party packages are stubbed,
`collections.Iterable` is aliased for Python 3.10, and `pyquaternion.Quaternion` (absent) is provided by an independent
implementation on scipy.spatial.transform.Rotation - only its `rotation_matrix` is used (nuscenes.py:45).
"""
import collections
import collections.abc
import importlib
import importlib.abc
import importlib.machinery
import json
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import ref_import as R  # noqa: E402

for _n in ("Iterable", "Mapping", "Sequence", "Callable"):
    if not hasattr(collections, _n):
        setattr(collections, _n, getattr(collections.abc, _n))

_PREFIXES = ("cv2", "tqdm", "nuscenes", "skimage", "fire", "tensorboardX", "apex", "matplotlib", "open3d", "easydict", "ipdb",
             "numba", "waymo_open_dataset", "tensorflow", "torchvision", "terminaltables", "spconv", "addict", "pycocotools",
             "filterpy", "shapely")


class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []

        def _ga(attr):
            if attr.startswith("__"):
                raise AttributeError(attr)
            return R._Permissive()

        m.__getattr__ = _ga
        return m

    def exec_module(self, m):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in _PREFIXES:
            return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)
        return None


def import_reference_dataset():
    from scipy.spatial.transform import Rotation
    sys.meta_path.append(_Finder())  # behind the real finders: only missing modules are stubbed
    pq = types.ModuleType("pyquaternion")

    class Quaternion:
        def __init__(self, q):
            self.q = np.asarray(q, dtype=float)

        @property
        def rotation_matrix(self):
            w, x, y, z = self.q
            return Rotation.from_quat([x, y, z, w]).as_matrix()

    pq.Quaternion = Quaternion
    sys.modules["pyquaternion"] = pq
    if R.REF_ROOT not in sys.path:
        sys.path.insert(0, R.REF_ROOT)
    return importlib.import_module("det3d.datasets.nuscenes.nuscenes")


NAMES = ["car", "pedestrian", "truck"]


def synth_frames(rng, n_frames, max_dets):
    """Inputs in the on-disk formats: token -> (13-float rows, class dicts), frame_info, labels."""
    tokens = ["tok%02d" % i for i in range(n_frames)]
    dets, cls, frame_info, labels = {}, {}, {}, {}
    for i, t in enumerate(tokens):
        k = int(rng.integers(0, max_dets + 1)) if i != 2 else 0  # frame 2 has no detection at all
        rows, cl = [], []
        for _ in range(k):
            q = rng.normal(size=4)
            q = q / np.linalg.norm(q) * float(rng.uniform(0.5, 2.0))  # not normalised on purpose
            rows.append([float(v) for v in np.concatenate([rng.uniform(-50, 50, 3), rng.uniform(0.5, 5, 3), q, rng.normal(size=2),
                                                           rng.uniform(0, 1, 1)])])
            cl.append(dict(detection_name=NAMES[int(rng.integers(0, 3))], detection_score=float(rng.uniform(0, 1)),
                           sample_token=t))
        dets[t], cls[t] = rows, cl
        frame_info[t] = dict(prev=tokens[i - 1] if i > 0 else "", timestamp=1_600_000_000_000_000 + 500_000 * i + int(rng.integers(0, 999)),
                             prev_timestamp=1_600_000_000_000_000 + 500_000 * (i - 1))
    for i, t in enumerate(tokens):
        k = len(dets[t])
        p = len(dets[tokens[i - 1]]) if i > 0 else 0
        matched = np.zeros((p, k + 2))
        free = list(range(k))
        for r in range(p):
            u = rng.uniform()
            if u < 0.55 and free:
                matched[r, free.pop(int(rng.integers(0, len(free))))] = 1  # matched to a current detection
            elif u < 0.8:
                matched[r, -2] = 1  # dead track
        newborn = np.zeros(k)
        for c in range(k):
            if matched[:, c].sum() == 0 and rng.uniform() < 0.5:
                newborn[c] = 1
        labels[t] = dict(matched=matched, newborn=newborn)
    return tokens, dets, cls, frame_info, labels


def write_tree(root, tokens, dets, cls, frame_info, labels):
    for d in ("det", "cls", "labels"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for t in tokens:
        with open(os.path.join(root, "det", t + ".json"), "w") as f:
            json.dump(dets[t], f)
        with open(os.path.join(root, "cls", t + ".json"), "w") as f:
            json.dump(cls[t], f)
        np.savez(os.path.join(root, "labels", t + ".npz"), matched=labels[t]["matched"], newborn=labels[t]["newborn"])
    with open(os.path.join(root, "frame_info.json"), "w") as f:
        json.dump(frame_info, f)


CASES = [dict(max_objects=6, det_type=None, fp_ratio=1.0, dead_trk_ratio=1.0, test_mode=False, seed=11),
         dict(max_objects=6, det_type=["car", "truck"], fp_ratio=0.5, dead_trk_ratio=0.5, test_mode=False, seed=12),
         dict(max_objects=4, det_type=None, fp_ratio=2.0, dead_trk_ratio=0.0, test_mode=False, seed=13),
         dict(max_objects=12, det_type=["pedestrian"], fp_ratio=1.0, dead_trk_ratio=1.0, test_mode=True, seed=14)]


def main():
    M = import_reference_dataset()
    rng = np.random.default_rng(2024)
    tokens, dets, cls, frame_info, labels = synth_frames(rng, 6, 9)
    out = dict(tokens=np.array(tokens), dets=np.array(json.dumps(dets)), cls=np.array(json.dumps(cls)),
               frame_info=np.array(json.dumps(frame_info)), cases=np.array(json.dumps(CASES)))
    for t in tokens:
        out["lab_matched_" + t] = labels[t]["matched"]
        out["lab_newborn_" + t] = labels[t]["newborn"]
    with tempfile.TemporaryDirectory() as root:
        write_tree(root, tokens, dets, cls, frame_info, labels)
        for ci, case in enumerate(CASES):
            infos = [dict(token=t) for t in tokens]
            fake = types.SimpleNamespace(
                _nusc_infos=infos, _frame_info=frame_info, _map_frame_token_idx={t: i for i, t in enumerate(tokens)},
                _max_objects=case["max_objects"], _det_path=os.path.join(root, "det"), _cls_info_path=os.path.join(root, "cls"),
                _det_type=case["det_type"], _labels_path=os.path.join(root, "labels"), test_mode=case["test_mode"],
                _dead_trk_ratio=case["dead_trk_ratio"], _fp_ratio=case["fp_ratio"], nsweeps=1, _root_path=root,
                _num_point_features=5, virtual=False, pipeline=lambda res, info: ({"metadata": res["metadata"]}, None))
            fake.get_frame_idx = lambda tok, _f=fake: M.NuScenesDataset.get_frame_idx(_f, tok)
            for i, t in enumerate(tokens):
                random.seed(case["seed"] * 100 + i)
                pre = "c%d_%s_" % (ci, t)
                try:
                    M.NuScenesDataset.get_sensor_data(fake, i)
                except IndexError:  # an empty frame in training mode: the reference indexes the label matrix out of range
                    out[pre + "raises"] = np.array([1])
                    continue
                info = infos[i]
                out[pre + "det_boxes"] = np.array(info["det_boxes"])
                out[pre + "prev_det_boxes"] = np.array(info["prev_det_boxes"])
                out[pre + "num"] = np.array([info["num_det_boxes"], info["num_prev_det_boxes"]])
                out[pre + "cls_scores"] = np.array([c["detection_score"] for c in info["cls_det_boxes"]])
                out[pre + "prev_cls_scores"] = np.array([c["detection_score"] for c in info["prev_cls_det_boxes"]])
                if not case["test_mode"]:
                    out[pre + "gt"] = np.array(info["gt"])
    path = os.path.join(HERE, "frames_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
