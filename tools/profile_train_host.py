"""GPU-box helper: where the HOST time of a training step goes (cProfile over a few steps of the car configuration).
usage: python tools/profile_train_host.py [--max-obj 90] [--points 5] [--feats 3] [--batch 64] [--steps 20]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402
from shasta_amd import training  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--max-obj", type=int, default=90)
ap.add_argument("--feats", type=int, default=3)
ap.add_argument("--points", type=int, default=5)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
with torch.device(dev):
    model = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                             bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                             max_obj=a.max_obj, num_feats=a.feats, num_point=a.points, in_channels=512)).train()
params = training.affinity_params(model)
opt = training.FusedAdam(params, lr=1e-4, lowrank_first_layers=model)
N, B = a.max_obj, a.batch
bev = torch.relu(torch.randn(B, 180, 180, 64, device=dev))
pbev = torch.relu(torch.randn(B, 180, 180, 64, device=dev))


def boxes():
    t = torch.zeros(B, N, 11, device=dev)
    t[:, :, :2] = (torch.rand(B, N, 2, device=dev) - 0.5) * 100
    t[:, :, 3:6] = torch.rand(B, N, 3, device=dev) * 3 + 0.5
    t[:, :, 6] = (torch.rand(B, N, device=dev) - 0.5) * 6.28
    return t


det0, prev0 = boxes(), boxes()
gt = (torch.rand(B, N + 2, N + 2, device=dev) < 0.02).float()
gt[:, 0, 0] = 1


def one():
    opt.zero_grad(set_to_none=True)
    m1, m2 = training.affinity_train(model, bev, pbev, det0.clone(), prev0.clone())
    training.affinity_loss(m1, m2, gt).backward()
    opt.step()


for _ in range(3):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    one()
t_host = (time.perf_counter() - t0) / a.steps * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / a.steps * 1e3
print("host time to issue a step %.3f ms, step incl. device %.3f ms" % (t_host, t_all))
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    one()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
# the backward runs on the autograd engine's thread: profile its body there
inner = cProfile.Profile()
orig = training._AffinityTrainFn.backward


def profiled(ctx, g1, g2):
    inner.enable()
    try:
        return orig(ctx, g1, g2)
    finally:
        inner.disable()


training._AffinityTrainFn.backward = staticmethod(profiled)
for _ in range(a.steps):
    one()
torch.cuda.synchronize()
pstats.Stats(inner).sort_stats("tottime").print_stats(25)
