"""GPU-box helper: time of the pair-residual stage (row embeddings + row_prep + pair kernel) per arithmetic.
    python tools/pair_time.py [--max-obj 90] [--points 5] [--feats 3] [--batch 512] [--iters 20] [--modes f16x2 pieces]
Run under `rocprofv3 --kernel-trace --stats` for the pair kernel alone."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402
from shasta_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--max-obj", type=int, default=90)
ap.add_argument("--points", type=int, default=5)
ap.add_argument("--feats", type=int, default=3)
ap.add_argument("--batch", type=int, nargs="+", default=[512])
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--modes", nargs="+", default=["f16x2", "pieces"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, F, T = a.max_obj, 64 * a.points, a.max_obj + 2
with torch.device(dev):
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
        bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
        max_obj=N, num_feats=a.feats, num_point=a.points)).eval()
lib = hip.load()
ld = (T + 3) // 4 * 4
for B in a.batch:
    feat, pfeat = torch.rand(B, T, F, device=dev), torch.rand(B, T, F, device=dev)
    dt, pt = torch.rand(B, T, 8, device=dev) * 4 + 0.5, torch.rand(B, T, 8, device=dev) * 4 + 0.5
    res = torch.empty(B, T, ld, device=dev)
    wsb = lib.shasta_forward_workspace_bytes(B, N, a.feats, F)
    ws = torch.empty(wsb // 4 + 1, device=dev)
    out = {}
    for mode in a.modes:
        m.arithmetic = mode
        w = m._weights()
        m._ensure_packed(w, dev)

        def run():
            hip.check(lib.shasta_pair_residual_f32(C.byref(w), hip.ptr(m._packed), B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(dt), hip.ptr(pt),
                                                   hip.ptr(res), ld, hip.ptr(ws), wsb, hip.stream_ptr()), "pair")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        out[mode] = (e0.elapsed_time(e1) / a.iters, res.clone())
    ref = out[a.modes[-1]][1]
    print(json.dumps(dict(max_obj=N, F=F, B=B, ms={k: round(v[0], 4) for k, v in out.items()},
                          max_abs_diff_vs_last={k: float((v[1][:, :, :T] - ref[:, :, :T]).abs().max()) for k, v in out.items()},
                          ref_scale=float(ref[:, :, :T].abs().max()))), flush=True)
