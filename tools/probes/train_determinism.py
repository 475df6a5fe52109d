"""GPU-box probe: is one training step from the neck outputs (car configuration) bit-reproducible?  Two runs from the same start in one
process: per-parameter gradients of step 0 compared bit for bit, then the first step at which the losses differ."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import shasta_amd  # noqa: E402
from shasta_amd import training  # noqa: E402

dev = torch.device("cuda:0")
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
gt[torch.arange(B)[:, None], torch.arange(N)[None, :], perm] = 1.0
gt = gt.to(dev)
runs = []
for rep in range(2):
    model = copy.deepcopy(base)
    opt = training.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    losses, grads0, outs0 = [], None, None
    for it in range(int(os.environ.get("STEPS", "30"))):
        opt.zero_grad(set_to_none=True)
        m1, m2, _ = model(dict(det_boxes=det0.clone(), prev_det_boxes=prev0.clone(), bev_map=x, prev_bev_map=xp), train_mode=True)
        loss = training.affinity_loss(m1, m2, gt)
        loss.backward()
        if it == 0:
            grads0 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            outs0 = (m1.detach().clone(), m2.detach().clone())
        opt.step()
        losses.append(loss.detach().clone())
    runs.append((torch.stack(losses).cpu(), grads0, outs0))
(la, ga, oa), (lb, gb, ob) = runs
print("forward of step 0 bit-identical:", torch.equal(oa[0], ob[0]) and torch.equal(oa[1], ob[1]))
for k in ga:
    if not torch.equal(ga[k], gb[k]):
        d = (ga[k] - gb[k]).abs().max().item()
        print("grad differs: %-28s max |diff| %.3e  (max |grad| %.3e)" % (k, d, ga[k].abs().max().item()))
diff = (la != lb).nonzero()
print("losses: first differing step", int(diff[0]) if len(diff) else None, "of", len(la))
