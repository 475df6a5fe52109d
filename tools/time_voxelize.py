"""GPU-box helper: time K1 (hard voxelisation + mean) on a nuScenes-sized cloud, next to the C oracle on the host."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import voxelize_oracle as VO  # noqa: E402
from shasta_amd.voxel_generator import points_to_voxel_batch_device, points_to_voxel_device  # noqa: E402

VS = np.array([0.075, 0.075, 0.2], np.float32)
RG = np.array([-54, -54, -5, 54, 54, 3], np.float32)
rng = np.random.default_rng(0)
for P in (100000, 300000):
    pts = np.zeros((P, 5), np.float32)
    r = np.abs(rng.normal(0, 18, size=P)).astype(np.float32)
    th = rng.uniform(0, 2 * np.pi, size=P).astype(np.float32)
    pts[:, 0], pts[:, 1] = r * np.cos(th), r * np.sin(th)
    pts[:, 2] = rng.normal(-1.5, 0.6, size=P)
    pts[:, 3] = rng.uniform(0, 255, size=P)
    d = torch.from_numpy(pts).cuda()
    for _ in range(3):
        out = points_to_voxel_device(d, VS, RG, 10, 160000, with_mean=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = points_to_voxel_device(d, VS, RG, 10, 160000, with_mean=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    VO.points_to_voxel(pts, VS, RG, 10, 160000)
    t0 = time.perf_counter()
    for _ in range(5):
        ref = VO.points_to_voxel(pts, VS, RG, 10, 160000, with_mean=True)
    dc = (time.perf_counter() - t0) / 5
    print("P=%d voxels=%d  HIP %.3f ms (incl. output allocation + one host sync)  C oracle %.3f ms" % (P, out[0].shape[0], dt * 1e3, dc * 1e3), flush=True)

# a batch: the current and the previous cloud of 8 samples (datasets/pipelines/preprocess.py:179-208 voxelises both) in one chain of
# launches, voxel counts left on the device
P, n = 300000, 16
clouds = []
for i in range(n):
    pts = np.zeros((P, 5), np.float32)
    r = np.abs(rng.normal(0, 18, size=P)).astype(np.float32)
    th = rng.uniform(0, 2 * np.pi, size=P).astype(np.float32)
    pts[:, 0], pts[:, 1] = r * np.cos(th), r * np.sin(th)
    pts[:, 2] = rng.normal(-1.5, 0.6, size=P)
    pts[:, 3] = rng.uniform(0, 255, size=P)
    clouds.append(pts)
allp = torch.from_numpy(np.concatenate(clouds)).cuda()
offs = [P * i for i in range(n + 1)]
for _ in range(3):
    out = points_to_voxel_batch_device((allp, offs), VS, RG, 10, 160000, with_mean=True)
torch.cuda.synchronize()
iters = int(os.environ.get("VOX_ITERS", "10"))
t0 = time.perf_counter()
for _ in range(iters):
    out = points_to_voxel_batch_device((allp, offs), VS, RG, 10, 160000, with_mean=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
V = int(out[4].sum())
alg = n * P * 5 * 4 + V * (10 * 5 * 4 + 3 * 4 + 4 + 5 * 4)
print("batch of %d clouds x %d points: %.3f ms per batch = %.0f clouds/s (incl. output allocation, no host read); %d voxels; algorithmic "
      "%.1f MB -> %.0f GB/s" % (n, P, dt * 1e3, n / dt, V, alg / 1e6, alg / dt / 1e9), flush=True)
