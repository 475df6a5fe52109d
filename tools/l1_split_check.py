"""First aug_shape layer (K3a: VALU, f32 MFMA or bf16-piece kernel): error of the kernel the library picks against a float64 evaluation of the same
relu(W x + b), and its duration.  Run once as is (bf16-piece kernel for batches > 32) and once with --f32 (Shasta.arithmetic = "f32": f32 MFMA):
the two error columns are what DESIGN.md section 4 quotes.
usage: python tools/l1_split_check.py [--max-obj 500] [--points 4] [--batch 64 128] [--steps 20] [--f32]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--max-obj", type=int, default=500)
ap.add_argument("--feats", type=int, default=7)
ap.add_argument("--points", type=int, default=4)
ap.add_argument("--batch", type=int, nargs="+", default=[64, 128])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--f32", action="store_true")
ap.add_argument("--arithmetic", choices=["pieces", "f32", "f16x2"], default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
with torch.device(dev):
    model = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                             bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                             max_obj=a.max_obj, num_feats=a.feats, num_point=a.points)).eval()
if a.f32:
    model.arithmetic = "f32"
if a.arithmetic:
    model.arithmetic = a.arithmetic
N = a.max_obj
K = model.aug_shape_input
for B in a.batch:
    g = torch.Generator(device=dev).manual_seed(B)
    bev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
    pbev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
    det0 = torch.zeros(B, N, 11, device=dev)
    det0[..., :2] = torch.rand(B, N, 2, device=dev, generator=g) * 100 - 50
    det0[..., 3:6] = torch.rand(B, N, 3, device=dev, generator=g) * 4 + 0.5
    det0[..., 6] = torch.rand(B, N, device=dev, generator=g) * 6.28 - 3.14
    det0[..., 9] = 0.5
    prev = det0.roll(1, 1).contiguous()
    keep = {}
    with torch.no_grad():
        model.affinity_from_bev(bev, pbev, det0.clone(), prev, _train_keep=keep)
        torch.cuda.synchronize()
        hid = keep["shape_hidden"].double()
        Hs = hid.shape[1] // 4
        err = scale = 0.0
        for i in range(4):
            x = keep["feat" if i < 2 else "prev_feat"].reshape(B, -1)[:, :K].double()
            lin = model.aug_shape[i][0]
            ref = torch.relu(x @ lin.weight.double().t() + lin.bias.double())
            err = max(err, (hid[:, i * Hs:(i + 1) * Hs] - ref).abs().max().item())
            scale = max(scale, ref.abs().max().item())
            del ref, x
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        det = det0.clone()
        for _ in range(3):
            det.copy_(det0)
            model.affinity_from_bev(bev, pbev, det, prev)
        t0.record()
        for _ in range(a.steps):
            det.copy_(det0)
            model.affinity_from_bev(bev, pbev, det, prev)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / a.steps
    print(json.dumps(dict(B=B, max_obj=N, K=K, f32_forced=bool(a.f32 or a.arithmetic == "f32"), arithmetic=model.arithmetic, max_abs_err=err, ref_scale=scale,
                          ms_per_step=round(ms, 4), frame_pairs_per_s=round(B / ms * 1e3, 1))), flush=True)
