"""GPU-box helper for profiler passes: nothing but K0 launches (one arithmetic, one batch).
    python tools/conv_only.py --batch 8 --iters 5 [--arith f16x2|f32] [--heads 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402
from shasta_amd.shared_conv import SharedConvBank  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--arith", default="f16x2")
ap.add_argument("--heads", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ms = []
for i in range(a.heads):
    torch.manual_seed(i)
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                         bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                         max_obj=4, num_feats=7, num_point=5)).eval().to(dev)
    m.arithmetic = a.arith
    ms.append(m)
x = torch.relu(torch.randn(a.batch, 512, 180, 180, device=dev))
xp = torch.relu(torch.randn(a.batch, 512, 180, 180, device=dev))
bank = SharedConvBank(ms) if a.heads > 1 else None
with torch.no_grad():
    for _ in range(a.iters):
        if bank is not None:
            bank(x, xp)
        else:
            ms[0].shared_conv_nhwc(x, xp)
torch.cuda.synchronize()
