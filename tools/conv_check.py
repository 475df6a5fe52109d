"""GPU-box helper: K0 (shared_conv) in every arithmetic against a float64 convolution on the device, and timed against MIOpen.

    python tools/conv_check.py [--heads 7] [--batches 1,2,8]
Prints one JSON line per measurement (kind = "accuracy" | "time")."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402
from shasta_amd.shared_conv import SharedConvBank  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--heads", type=int, default=7)
ap.add_argument("--batches", default="1,2,8")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda", 0)


def model(seed, cin=512):
    torch.manual_seed(seed)
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                         bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                         max_obj=4, num_feats=7, num_point=5, in_channels=cin)).eval()
    with torch.no_grad():
        m.shared_conv[1].running_mean.copy_(torch.randn(64) * 0.3)
        m.shared_conv[1].running_var.copy_(torch.rand(64) + 0.5)
        m.shared_conv[1].weight.copy_(torch.rand(64) + 0.5)
        m.shared_conv[1].bias.copy_(torch.randn(64) * 0.2)
    return m.to(dev)


def ref64(m, x):
    conv, bn = m.shared_conv[0], m.shared_conv[1]
    y = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    y = (y - bn.running_mean.double()[None, :, None, None]) / torch.sqrt(bn.running_var.double() + bn.eps)[None, :, None, None]
    y = y * bn.weight.double()[None, :, None, None] + bn.bias.double()[None, :, None, None]
    return torch.relu(y).permute(0, 2, 3, 1).contiguous()


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


models = [model(10 + i) for i in range(args.heads)]
m = models[0]
with torch.no_grad():
    # accuracy: random maps (relu of normal, one map scaled by 1e-3, one by 1e3), the real shape and ragged ones
    for (B, cin, H, W, scale) in [(1, 512, 180, 180, 1.0), (2, 512, 180, 180, 1e-3), (1, 512, 180, 180, 1e3), (2, 32, 33, 70, 1.0), (1, 16, 7, 45, 1.0),
                                  (3, 48, 5, 187, 1.0), (1, 16, 300, 2, 1.0)]:
        mm = m if cin == 512 else model(99, cin)
        g = torch.Generator(device=dev).manual_seed(5)
        x = torch.relu(torch.randn(B, cin, H, W, device=dev, generator=g)) * scale
        xp = torch.relu(torch.randn(B, cin, H, W, device=dev, generator=g)) * scale
        r, rp = ref64(mm, x), ref64(mm, xp)
        den = float(r.abs().max())
        out = {"kind": "accuracy", "shape": [B, cin, H, W], "scale": scale, "ref_max": den}
        for arith in ("f32", "f16x2"):
            mm.arithmetic = arith
            y, yp = mm.shared_conv_nhwc(x, xp)
            out[arith] = max(float((y.double() - r).abs().max()), float((yp.double() - rp).abs().max())) / den
        mi = mm.shared_conv(x).permute(0, 2, 3, 1)
        out["miopen"] = float((mi.double() - r).abs().max()) / den
        print(json.dumps(out), flush=True)
    # multi-head equals single-head bit for bit
    bank = SharedConvBank(models)
    x = torch.relu(torch.randn(1, 512, 180, 180, device=dev))
    xp = torch.relu(torch.randn(1, 512, 180, 180, device=dev))
    outs, outs_p = bank(x, xp)
    same = True
    for i, mm in enumerate(models):
        mm.arithmetic = "f16x2"
        y, yp = mm.shared_conv_nhwc(x, xp)
        same = same and torch.equal(y, outs[i]) and torch.equal(yp, outs_p[i])
    print(json.dumps({"kind": "multi_equals_single", "heads": args.heads, "equal": bool(same)}), flush=True)

    flop_map = 2 * 180 * 180 * 64 * 4608
    for B in [int(b) for b in args.batches.split(",")]:
        x = torch.relu(torch.randn(B, 512, 180, 180, device=dev))
        xp = torch.relu(torch.randn(B, 512, 180, 180, device=dev))
        for name, fn, nmap in (
                ("f32", lambda: (setattr(m, "arithmetic", "f32"), m.shared_conv_nhwc(x, xp)), 2 * B),
                ("f16x2", lambda: (setattr(m, "arithmetic", "f16x2"), m.shared_conv_nhwc(x, xp)), 2 * B),
                ("f16x2_heads%d" % args.heads, lambda: bank(x, xp), 2 * B * args.heads),
                ("miopen", lambda: (m.shared_conv(x).permute(0, 2, 3, 1).contiguous(), m.shared_conv(xp).permute(0, 2, 3, 1).contiguous()), 2 * B)):
            dt = timed(fn, args.iters)
            print(json.dumps({"kind": "time", "B": B, "name": name, "ms": dt * 1e3, "ms_per_frame_pair_per_head": dt * 1e3 / (nmap / 2),
                              "tflops_fp32_equiv": nmap * flop_map / dt / 1e12}), flush=True)
