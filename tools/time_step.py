"""GPU-box helper: ms per forward step (inference path from BEV maps) for the library in SHASTA_HIP_LIB, no result checks (probe builds
may compute garbage).  usage: time_step.py B [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
import shasta_amd  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
with torch.device(dev):
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
        bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
        max_obj=500, num_feats=7, num_point=4)).eval()
Bs = [int(x) for x in sys.argv[1:]] or [512]
Bmax = max(Bs)
g = torch.Generator(device=dev).manual_seed(1)
bev = torch.relu(torch.randn(Bmax, 180, 180, 64, device=dev, generator=g))
pbev = torch.relu(torch.randn(Bmax, 180, 180, 64, device=dev, generator=g))
N = 500


def boxes():
    b = torch.zeros(Bmax, N, 11, device=dev)
    b[..., 0:2] = torch.rand(Bmax, N, 2, device=dev, generator=g) * 100 - 50
    b[..., 2] = torch.randn(Bmax, N, device=dev, generator=g)
    b[..., 3:6] = torch.rand(Bmax, N, 3, device=dev, generator=g) * 4 + 0.5
    b[..., 6] = (torch.rand(Bmax, N, device=dev, generator=g) * 2 - 1) * 3.14159265
    b[..., 7:9] = torch.randn(Bmax, N, 2, device=dev, generator=g)
    b[..., 9] = 0.5
    return b


det, prev = boxes(), boxes()
out = []
for B in Bs:
    d = det[:B].clone()
    with torch.no_grad():
        for _ in range(10):
            m.affinity_from_bev(bev[:B], pbev[:B], d, prev[:B])
        torch.cuda.synchronize()
        n = max(20, int(600 / (B * 0.018 + 0.7)))
        t0 = time.perf_counter()
        for _ in range(n):
            m.affinity_from_bev(bev[:B], pbev[:B], d, prev[:B])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    out.append("B=%d %.3f ms (%.0f fp/s)" % (B, ms, B / ms * 1e3))
print(os.path.basename(os.environ.get("SHASTA_HIP_LIB", "default")), " | ".join(out))
