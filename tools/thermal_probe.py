"""GPU-box helper: how long does the pair-residual stage take when the chip was idle before it?  (rocprofv3 --pmc serialises the
kernels with idle gaps in between and shows pair_f16_kernel at 3.0 - 3.2 ms, the free-running step 4.2 - 4.5 ms.)
usage: thermal_probe.py [B]"""
import ctypes as C
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
import shasta_amd  # noqa: E402
from shasta_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with torch.device(dev):
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
        bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
        max_obj=500, num_feats=7, num_point=4)).eval()
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


def call():
    hip.check(lib.shasta_pair_residual_f32(C.byref(w), hip.ptr(m._packed), B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(dt),
                                           hip.ptr(pt), hip.ptr(res), 504, hip.ptr(ws), wsb, hip.stream_ptr()), "pair")


def timed(n=1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        call()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


for _ in range(20):
    call()
torch.cuda.synchronize()
print("back to back, 200 calls: %.3f ms per call" % timed(200))
for gap in (0.0, 0.001, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.3):
    ts = []
    for _ in range(15):
        torch.cuda.synchronize()
        time.sleep(gap)
        ts.append(timed(1))
    print("idle %5.0f ms before one call: median %.3f ms (min %.3f max %.3f)" % (gap * 1e3, statistics.median(ts), min(ts), max(ts)))
series = []
for _ in range(7):
    torch.cuda.synchronize()
    time.sleep(0.3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for i in range(40):
        call()
        ev[i + 1].record()
    ev[-1].synchronize()
    series.append([ev[i].elapsed_time(ev[i + 1]) for i in range(40)])
med = [statistics.median(s[i] for s in series) for i in range(40)]
print("40 consecutive calls after 300 ms idle (median of 7): " + " ".join("%.2f" % x for x in med))
print("back to back again: %.3f ms per call" % timed(200))
