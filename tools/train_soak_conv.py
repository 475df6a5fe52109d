"""GPU-box helper: 200 training steps of the car configuration FROM THE NECK OUTPUTS (512 x 180 x 180 maps, shared_conv.0 / .1 trained
with everything else: tools/nusc_shasta/train.py:186-218) with K0 hand-written (shared_conv_train.hip) and, from the same start, through
nn.Sequential (MIOpen / ATen): the loss curves side by side, finite parameters, ms per step.  usage: python tools/train_soak_conv.py"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402
from shasta_amd import training  # noqa: E402

dev = torch.device("cuda:0")
STEPS = int(os.environ.get("SOAK_STEPS", "200"))
torch.manual_seed(0)
base = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                        bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                        max_obj=90, num_feats=3, num_point=5, in_channels=512)).to(dev).train()
g = torch.Generator(device="cpu").manual_seed(1)
B, N = 4, 90
x = torch.relu(torch.randn(B, 512, 180, 180, generator=g)).to(dev)
xp = torch.relu(torch.randn(B, 512, 180, 180, generator=g)).to(dev)


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
gt = torch.zeros(B, N + 2, N + 2)
perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
gt[torch.arange(B)[:, None], torch.arange(N)[None, :], perm] = 1.0  # a fixed permutation per frame pair: learnable
gt = gt.to(dev)
curves = {}
for name in ("hand-written K0", "nn.Sequential K0"):
    model = copy.deepcopy(base)
    model.hand_written_train_conv = name.startswith("hand")
    opt = training.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=3e-4, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=STEPS)
    losses = []
    for it in range(STEPS):
        if it == 20:  # the clock starts behind MIOpen's first-call search
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        m1, m2, _ = model(dict(det_boxes=det0.clone(), prev_det_boxes=prev0.clone(), bev_map=x, prev_bev_map=xp), train_mode=True)
        loss = training.affinity_loss(m1, m2, gt)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.detach())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (STEPS - 20)
    losses = [float(l) for l in losses]
    finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    used = getattr(model, "_conv_raw", None) is not None
    curves[name] = losses
    print("%-17s %.2f ms/step (steps 20 on)  finite=%s  hand-written kernels ran=%s  loss %s" % (
        name, dt * 1e3, finite, used, " ".join("%d:%.5f" % (i, losses[i]) for i in (0, 1, 2, 5, 10, 20, 50, 100, STEPS - 1) if i < STEPS)), flush=True)
a, b = curves["hand-written K0"], curves["nn.Sequential K0"]
rel = [abs(p - q) / max(abs(q), 1e-12) for p, q in zip(a, b)]
print("relative difference of the two loss curves: step 0 %.2e, max over steps 0-9 %.2e, 10-49 %.2e, 50-%d %.2e" % (
    rel[0], max(rel[:10]), max(rel[10:50]), STEPS - 1, max(rel[50:])), flush=True)
