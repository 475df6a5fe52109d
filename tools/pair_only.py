"""GPU-box helper: run only the pair-residual stage (for rocprofv3 --pmc passes).  usage: pair_only.py B iters"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import shasta_amd  # noqa: E402
from shasta_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with torch.device(dev):
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
        bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
        max_obj=500, num_feats=7, num_point=4)).eval()
    # the 4 GB anchor matrices are not needed here: shrink them so the process starts quickly
lib = hip.load()
w = m._weights()
m._ensure_packed(w, dev)
N, F, T = 500, 256, 502
feat = torch.rand(B, T, F, device=dev)
pfeat = torch.rand(B, T, F, device=dev)
dt = torch.rand(B, T, 8, device=dev) * 4 + 0.5
pt = torch.rand(B, T, 8, device=dev) * 4 + 0.5
res = torch.empty(B, T, 504, device=dev)
wsb = lib.shasta_forward_workspace_bytes(B, N, 7, F)
ws = torch.empty(wsb // 4 + 1, device=dev)
for _ in range(iters):
    hip.check(lib.shasta_pair_residual_f32(C.byref(w), hip.ptr(m._packed), B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(dt),
                                           hip.ptr(pt), hip.ptr(res), 504, hip.ptr(ws), wsb, hip.stream_ptr()), "pair")
torch.cuda.synchronize()
print("done")
