"""Synthetic stand-in for the preprocessed nuScenes split (BASELINE configs 2-4): writes the SAME on-disk files
the reference's preprocessing/ writes and its dataset reads, so that the whole inference chain - loader -> batched affinity forward ->
decode -> `cp_<split>.json` -> merge -> tracker -> `tracking_result.json` - runs without the nuScenes data set or the devkit.

  <root>/<split>_2hz/detections/<det>/sensor_individual_frames/<token>.json   13-float rows [t(3), wlh(3), quat wxyz(4), vxy(2), score]
                                                                              (preprocessing/get_det_sensor_info.py:98-106)
  <root>/<split>_2hz/detections/<det>/cls_individual_frames/<token>.json      nuScenes detection dicts (preprocessing/get_det_info.py:44-46)
  <root>/<split>_frame_info.json                                             {token: {prev, next, timestamp, prev_timestamp, next_timestamp}}
                                                                              (preprocessing/get_frame_info.py:43-44; timestamps in microseconds)
  <root>/frames_meta.json                                                    {'frames': [{token, timestamp (s), first}]}
                                                                              (tools/nusc_shasta/eval.py:197-223 `save_first_frame`)

Objects move with constant velocity; detections drop out, clutter appears; every tracking class is present with its own object count.
There is no LiDAR here: the BEV feature maps that the (out-of-scope) backbone would produce are a deterministic function of the
frame token (`TokenBev`)."""
import hashlib
import json
import math
import os

import numpy as np

TRACKING_NAMES = ["bicycle", "bus", "car", "motorcycle", "pedestrian", "trailer", "truck"]
# objects per scene and class (roughly the nuScenes proportions, small enough for test-sized tables)
DEFAULT_OBJECTS = {"car": 14, "pedestrian": 10, "truck": 5, "trailer": 2, "bus": 2, "bicycle": 3, "motorcycle": 3}
CLASS_SIZE = {"car": (1.9, 4.6, 1.7), "truck": (2.5, 6.9, 2.8), "bus": (2.9, 11.0, 3.4), "trailer": (2.9, 12.0, 3.8),
              "pedestrian": (0.7, 0.7, 1.8), "bicycle": (0.6, 1.7, 1.3), "motorcycle": (0.8, 2.1, 1.5)}


def split_paths(root, split="val", det="cp"):
    d = os.path.join(root, "%s_2hz" % split, "detections", det)
    return dict(det_path=os.path.join(d, "sensor_individual_frames"), cls_info_path=os.path.join(d, "cls_individual_frames"),
                frame_info_path=os.path.join(root, "%s_frame_info.json" % split), frames_meta_path=os.path.join(root, "frames_meta.json"))


def write_synthetic_split(root, n_scenes=3, frames_per_scene=6, seed=0, split="val", det="cp", objects=None, dt=0.5,
                          drop=0.12, clutter=2):
    """Writes the files listed above; returns (paths dict, scenes: list of (scene name, [tokens in time order])).
    frames_per_scene: one length for every scene, or one length per scene (nuScenes scenes are about 40 key frames, not all equal)."""
    rng = np.random.default_rng(seed)
    lengths = [int(frames_per_scene)] * n_scenes if np.isscalar(frames_per_scene) else [int(v) for v in frames_per_scene]
    assert len(lengths) == n_scenes and min(lengths) >= 1
    objects = dict(DEFAULT_OBJECTS if objects is None else objects)
    p = split_paths(root, split, det)
    os.makedirs(p["det_path"], exist_ok=True)
    os.makedirs(p["cls_info_path"], exist_ok=True)
    frame_info, frames_meta, scenes = {}, [], []
    t0 = 1_500_000_000_000_000
    for s in range(n_scenes):
        tokens = [hashlib.md5(("%d/%d/%d" % (seed, s, f)).encode()).hexdigest() for f in range(lengths[s])]
        scenes.append(("scene-%04d" % s, tokens))
        objs = []
        for name, n in objects.items():
            for _ in range(n):
                w, l, h = CLASS_SIZE[name]
                spd = {"pedestrian": 1.2, "bicycle": 3.0}.get(name, 6.0)
                objs.append(dict(name=name, pos=rng.uniform(-45, 45, 2), z=float(rng.normal(-1.0, 0.3)), vel=rng.normal(0, spd / 2, 2),
                                 size=[float(w * rng.uniform(0.9, 1.1)), float(l * rng.uniform(0.9, 1.1)), float(h * rng.uniform(0.9, 1.1))],
                                 yaw=float(rng.uniform(-math.pi, math.pi))))
        for f, tok in enumerate(tokens):
            ts = t0 + int((s * 1000 + f * dt) * 1e6)
            rows, cls = [], []
            for o in objs:
                if rng.uniform() < drop:
                    continue
                pos = o["pos"] + o["vel"] * dt * f + rng.normal(0, 0.1, 2)
                if abs(pos[0]) > 53 or abs(pos[1]) > 53:
                    continue
                yaw = o["yaw"] + float(rng.normal(0, 0.03))
                quat = [math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)]
                vel = (o["vel"] + rng.normal(0, 0.2, 2)).tolist()
                score = float(rng.uniform(0.2, 0.95))
                rows.append([float(pos[0]), float(pos[1]), o["z"]] + o["size"] + quat + vel + [score])
                cls.append(dict(sample_token=tok, translation=[float(pos[0]), float(pos[1]), o["z"]], size=o["size"], rotation=quat,
                                velocity=vel, detection_name=o["name"], detection_score=score, attribute_name=""))
            for _ in range(int(rng.integers(0, clutter + 1))):
                name = TRACKING_NAMES[int(rng.integers(0, len(TRACKING_NAMES)))]
                pos, yaw = rng.uniform(-50, 50, 2), float(rng.uniform(-math.pi, math.pi))
                quat = [math.cos(yaw / 2), 0.0, 0.0, math.sin(yaw / 2)]
                vel, score = rng.normal(0, 1, 2).tolist(), float(rng.uniform(0.05, 0.4))
                size = [float(v) for v in CLASS_SIZE[name]]
                rows.append([float(pos[0]), float(pos[1]), -1.0] + size + quat + vel + [score])
                cls.append(dict(sample_token=tok, translation=[float(pos[0]), float(pos[1]), -1.0], size=size, rotation=quat, velocity=vel,
                                detection_name=name, detection_score=score, attribute_name=""))
            order = rng.permutation(len(rows))
            with open(os.path.join(p["det_path"], tok + ".json"), "w") as fh:
                json.dump([rows[i] for i in order], fh)
            with open(os.path.join(p["cls_info_path"], tok + ".json"), "w") as fh:
                json.dump([cls[i] for i in order], fh)
            prev_tok, next_tok = (tokens[f - 1] if f else ""), (tokens[f + 1] if f + 1 < len(tokens) else "")
            frame_info[tok] = dict(prev=prev_tok, next=next_tok, timestamp=ts, prev_timestamp=ts - int(dt * 1e6) if f else ts,
                                   next_timestamp=ts + int(dt * 1e6) if next_tok else ts)
            frames_meta.append(dict(token=tok, timestamp=ts * 1e-6, first=(f == 0)))
    with open(p["frame_info_path"], "w") as fh:
        json.dump(frame_info, fh)
    with open(p["frames_meta_path"], "w") as fh:
        json.dump({"frames": frames_meta}, fh)
    return p, scenes


class TokenBev:
    """Stand-in for `shared_conv(neck(backbone(voxels)))` of a frame (shasta.py:223-228): an (H, W, C) NHWC fp32 map that is a
    deterministic function of the frame token (seeded CPU generator, so the device path and the CPU oracle see the same bytes)."""

    def __init__(self, hw=180, channels=64, seed=0):
        self.hw, self.channels, self.seed = hw, channels, seed
        self._cache = {}

    def __call__(self, token):
        import torch
        if token not in self._cache:
            if len(self._cache) > 64:
                self._cache.clear()
            g = torch.Generator().manual_seed((int(hashlib.md5(token.encode()).hexdigest()[:12], 16) + self.seed) % (2 ** 31))
            self._cache[token] = torch.relu(torch.randn(self.hw, self.hw, self.channels, generator=g))
        return self._cache[token]


class TokenNeck:
    """Stand-in for `neck(backbone(voxels))` of a frame (the (Cin, H, W) map shared_conv reads, shasta.py:223): a deterministic function
    of the frame token, generated ON the device (a seeded device generator per token), so that a timed chain holds no host-side
    stand-in work.  `neck_batch(tokens, device)` -> (n, Cin, H, W) fp32."""

    # relu of standard normals drawn from 24-bit uniforms: never above sqrt(2 ln 2^24) = 5.8.  A real neck knows the range of its own
    # ReLU(BatchNorm(.)) output the same way; shared_conv then skips its pass over the maps (SharedConvBank(..., bound=))
    neck_bound = 8.0

    def __init__(self, hw=180, channels=512, seed=0):
        self.hw, self.channels, self.seed = hw, channels, seed
        self._gen = {}

    def neck_batch(self, tokens, device):
        import torch
        g = self._gen.setdefault(str(device), torch.Generator(device=device))
        out = torch.empty(len(tokens), self.channels, self.hw, self.hw, device=device)
        for i, t in enumerate(tokens):
            g.manual_seed((int(hashlib.md5(t.encode()).hexdigest()[:12], 16) + self.seed) % (2 ** 31))
            out[i].normal_(generator=g)
        return torch.relu_(out)
