"""Times one training step (forward + HIP backward + Adam) of the affinity network on synthetic data.
usage: python tools/time_train.py [--max-obj 90] [--feats 3] [--points 5] [--batch 4] [--steps 5]"""
import argparse
import os
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
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--hw", type=int, default=180)
ap.add_argument("--json", action="store_true", help="print one JSON line instead of text")
ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused HIP step")
ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32", help="Shasta.train_precision: operands of the pair / aff GEMMs of the backward")
ap.add_argument("--in-backward", action="store_true", help="FusedAdam(in_backward=True): the first aug_shape layers step inside loss.backward()")
ap.add_argument("--graph", action="store_true", help="the whole step captured into a hipGraph and replayed (training.GraphedTrainStep, FusedAdam(capturable=True))")
a = ap.parse_args()
rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", device_id=dev)
torch.manual_seed(0)  # identical initial weights on every rank
cfg = dict(type="Shasta", reader=None, backbone=None, neck=None,
           bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
           max_obj=a.max_obj, num_feats=a.feats, num_point=a.points, in_channels=512)
model = shasta_amd.build_simp_track(cfg).to(dev).train()
model.train_precision = a.precision
params = training.affinity_params(model)
opt = torch.optim.Adam(params, lr=1e-4) if a.torch_adam else training.FusedAdam(params, lr=1e-4, lowrank_first_layers=model, in_backward=a.in_backward,
                                                                                capturable=a.graph)
N, B = a.max_obj, a.batch
g = torch.Generator(device="cpu").manual_seed(1 + rank)
bev = torch.relu(torch.randn(B, a.hw, a.hw, 64, generator=g)).to(dev)
pbev = torch.relu(torch.randn(B, a.hw, a.hw, 64, generator=g)).to(dev)


def boxes():
    t = torch.zeros(B, N, 11)
    t[:, :, :2] = (torch.rand(B, N, 2, generator=g) - 0.5) * 100
    t[:, :, 2] = torch.randn(B, N, generator=g)
    t[:, :, 3:6] = torch.rand(B, N, 3, generator=g) * 3 + 0.5
    t[:, :, 6] = (torch.rand(B, N, generator=g) - 0.5) * 6.28
    t[:, :, 7:9] = torch.randn(B, N, 2, generator=g)
    t[:, :, 9] = 0.5
    return t.to(dev)


det0, prev0 = boxes(), boxes()
gt = (torch.rand(B, N + 2, N + 2, generator=g) < 0.02).float().to(dev)
gt[:, 0, 0] = 1
if a.graph:
    step = training.GraphedTrainStep(model, opt, bev, pbev, det0, prev0, gt)
    for it in range(a.steps + 2):
        if it == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = step(bev, pbev, det0, prev0, gt)
for it in range(0 if a.graph else a.steps + 2):
    if it == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    m1, m2 = training.affinity_train(model, bev, pbev, det0.clone(), prev0.clone())
    loss = training.affinity_loss(m1, m2, gt)
    loss.backward()
    training.allreduce_gradients(params)  # no-op on one rank
    opt.step()
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = (time.perf_counter() - t0) / a.steps
if world > 1:
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
if rank == 0 and a.json:
    import json
    print(json.dumps({"metric": "affinity-net training frame-pairs/sec", "value": world * B / dt, "unit": "frame-pairs/s", "n_gpus": world,
                      "steps": a.steps, "ms_per_step": dt * 1e3, "dtype": "f32", "data": "synthetic", "scaling": "weak",
                      "config": {"workload": "synthetic training step (forward + HIP backward + factor exchange + fused Adam)",
                                 "max_obj": N, "feat_dim": model.aug_shape_output, "num_feats": a.feats,
                                 "frame_pairs_per_step_per_gpu": B, "optimizer": "torch Adam" if a.torch_adam else "FusedAdam"},
                      "loss": float(loss.detach()), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30}))
elif rank == 0:
    print("max_obj=%d F=%d B=%d: %.2f ms/step, %.1f frame-pairs/s, loss %.4f, peak mem %.2f GB" % (
        N, model.aug_shape_output, B, dt * 1e3, world * B / dt, float(loss.detach()), torch.cuda.max_memory_allocated() / 2**30))
if world > 1:
    dist.destroy_process_group()
