"""K0 in train() mode, timed: forward (conv + batch-statistics BatchNorm + ReLU -> NHWC, both maps of B frame pairs) and backward
(dgamma, dbeta, dbias, dweight) of Shasta.shared_conv_nhwc, hand-written (csrc/shared_conv_train.hip) against the module's own
nn.Sequential (MIOpen conv / wgrad, ATen BatchNorm, autograd).  usage: python tools/time_conv_train.py [--batch 8] [--iters 10] [--only hand|torch]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--cin", type=int, default=512)
ap.add_argument("--hw", type=int, default=180)
ap.add_argument("--only", choices=["hand", "torch", "both"], default="both")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                         bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                         max_obj=4, num_feats=3, num_point=5, in_channels=a.cin)).to(dev).train()
g = torch.Generator(device=dev).manual_seed(1)
B = a.batch
x = torch.relu(torch.randn(B, a.cin, a.hw, a.hw, device=dev, generator=g))
xp = torch.relu(torch.randn(B, a.cin, a.hw, a.hw, device=dev, generator=g))
go = torch.randn(B, a.hw, a.hw, 64, device=dev, generator=g) * (torch.rand(B, a.hw, a.hw, 1, device=dev, generator=g) < 0.06)
out = {"batch": B, "cin": a.cin, "hw": a.hw, "gflop_fwd": 2 * 2 * B * a.hw * a.hw * 64 * a.cin * 9 / 1e9}
for name, hand in (("hand_written", True), ("nn_sequential", False)):
    if a.only != "both" and (a.only == "hand") != hand:
        continue
    model.hand_written_train_conv = hand
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.iters)]
    for it in range(a.iters + 2):
        model.zero_grad(set_to_none=True)
        e = ev[it - 2] if it >= 2 else None
        if e:
            e[0].record()
        o, op = model.shared_conv_nhwc(x, xp)
        if e:
            e[1].record()
        torch.autograd.backward([o, op], [go, go])
        if e:
            e[2].record()
    torch.cuda.synchronize()
    fwd = sorted(e[0].elapsed_time(e[1]) for e in ev)[len(ev) // 2]
    bwd = sorted(e[1].elapsed_time(e[2]) for e in ev)[len(ev) // 2]
    out[name] = {"fwd_ms": fwd, "bwd_ms": bwd, "step_ms": fwd + bwd, "fwd_tflops": out["gflop_fwd"] / fwd, "wgrad_tflops": out["gflop_fwd"] / bwd}
print(json.dumps(out))
