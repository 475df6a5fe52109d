"""GPU-box helper: time the pair-residual stage alone for several tracks-per-workgroup settings (SHASTA_PAIR_TT)."""
import os
import subprocess
import sys

CODE = r'''
import os, sys, time, ctypes as C, torch
sys.path.insert(0, os.getcwd())
import shasta_amd
from shasta_amd import hip
B = int(sys.argv[1])
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with torch.device(dev):
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
        bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
        max_obj=500, num_feats=7, num_point=4)).eval()
lib = hip.load()
w = m._weights(); m._ensure_packed(w, dev)
N, F, T = 500, 256, 502
feat = torch.rand(B, T, F, device=dev); pfeat = torch.rand(B, T, F, device=dev)
dt = torch.rand(B, T, 8, device=dev) * 4 + 0.5; pt = torch.rand(B, T, 8, device=dev) * 4 + 0.5
res = torch.empty(B, T, 504, device=dev)
wsb = lib.shasta_forward_workspace_bytes(B, N, 7, F); ws = torch.empty(wsb // 4 + 1, device=dev)
def run():
    hip.check(lib.shasta_pair_residual_f32(C.byref(w), hip.ptr(m._packed), B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(dt),
              hip.ptr(pt), hip.ptr(res), 504, hip.ptr(ws), wsb, hip.stream_ptr()), "pair")
for _ in range(3): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 20
print("B=%d stagger=%s TT=%s pair stage %.1f us (%.2f us per frame-pair)" % (B, os.environ.get("SHASTA_PAIR_STAGGER", "0"), os.environ.get("SHASTA_PAIR_TT", "auto"), dt_ * 1e6, dt_ * 1e6 / B))
'''
for B in (1, 8, 32, 64):
    for mf in (0, 1, 2):
        env = dict(os.environ)
        if mf == 1:
            env["SHASTA_PAIR_MFMA"] = "1"
        if mf == 2:
            env["SHASTA_PAIR_VALU"] = "1"
        r = subprocess.run([sys.executable, "-c", CODE, str(B)], env=env, capture_output=True, text=True)
        print(("mfma4x4   ", "mfma16-chain", "valu-pk  ")[mf] + " "  + (r.stdout.strip() or r.stderr[-800:]), flush=True)
