"""Pair stage (K4: fuse_shape / fuse_det / res_coeff tails + hand residual + combine): error of the kernel the library picks against a
float64 evaluation of the reference formulation on the SAME feature / box tables, for the arithmetic modes.
usage: python tools/pair_check.py [--max-obj 500] [--points 4] [--feats 7] [--batch 2]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402
from oracle import shasta_oracle as O  # noqa: E402  (checker only)

ap = argparse.ArgumentParser()
ap.add_argument("--max-obj", type=int, default=500)
ap.add_argument("--feats", type=int, default=7)
ap.add_argument("--points", type=int, default=4)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--gain", type=float, default=1.0, help="scale of the pair-MLP weight matrices (sharpened weights)")
ap.add_argument("--spread", type=float, default=0.0, help="the BEV maps are scaled by 10^(+-spread) across their width: table rows of very "
                                                          "different magnitude meet in one detection tile (range scaling of the fp16 form)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
with torch.device(dev):
    model = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                             bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                             max_obj=a.max_obj, num_feats=a.feats, num_point=a.points)).eval()
if a.gain != 1.0:
    with torch.no_grad():
        for m in (model.fuse_shape, model.fuse_det, model.res_coeff):
            for l in m:
                if hasattr(l, "weight"):
                    l.weight.mul_(a.gain)
N, B = a.max_obj, a.batch
g = torch.Generator(device=dev).manual_seed(3)
bev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
pbev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
if a.spread:
    ramp = torch.pow(10.0, torch.linspace(-a.spread, a.spread, 180, device=dev)).view(1, 1, 180, 1)
    bev, pbev = bev * ramp, pbev * ramp
gc = torch.Generator().manual_seed(4)
det0, prev = O.synth_boxes(gc, B, N).to(dev), O.synth_boxes(gc, B, N).to(dev)
model.keep_intermediates = True
ref = None
for mode in ("f16x2", "pieces", "f32", "f16grid"):
    model.arithmetic = mode
    with torch.no_grad():
        model.affinity_from_bev(bev, pbev, det0.clone(), prev)
    torch.cuda.synchronize()
    im = {k: v.double().cpu() for k, v in model.last_intermediates.items()}
    if ref is None:  # float64 evaluation of shasta.py:277-319 on the tables the device produced
        w64 = {k: v.detach().double().cpu() for k, v in model.state_dict().items()}
        ref = O.pair_residual(w64, im["prev_feature"], im["feature"], im["prev_tab"][:, :, :7], im["det_tab"][:, :, :7], a.feats, chunk=16)
    err = (im["residual"] - ref).abs()
    print(json.dumps(dict(arithmetic=mode, max_obj=N, F=64 * a.points, max_abs_err=float(err.max()), ref_scale=float(ref.abs().max()),
                          rms_err=float(err.pow(2).mean().sqrt()))), flush=True)
