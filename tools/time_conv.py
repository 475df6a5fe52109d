"""GPU-box helper: time K0 (shared_conv HIP kernel) against the MIOpen path of nn.Conv2d + BatchNorm2d + ReLU."""
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
                                         bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54],
                                                            voxel_size=[0.075, 0.075], out_stride=8),
                                         max_obj=4, num_feats=7, num_point=5)).eval()
for B in (1, 2, 8):
    x = torch.relu(torch.randn(B, 512, 180, 180, device=dev))
    with torch.no_grad():
        for name, fn in (("hip ", m.shared_conv_nhwc), ("hip2", lambda t: m.shared_conv_nhwc(t, t)), ("miopen", lambda t: m.shared_conv(t).permute(0, 2, 3, 1).contiguous())):
            for _ in range(3):
                y = fn(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                y = fn(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            nmap = 2 * B if name == "hip2" else B
            print("B=%d %s %.3f ms  (%.1f TFLOP/s)" % (B, name, dt * 1e3, nmap * 2 * 180 * 180 * 64 * 4608 / dt / 1e12), flush=True)
