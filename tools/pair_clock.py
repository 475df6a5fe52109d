"""GPU-box helper: the clock pair_f16_kernel really runs at (MI355X_MICROARCH.md, DVFS give-back item 6).  Needs the diagnostic build
   python tools/build_variant.py pstamp pair_f16.hip -DPAIR_STAMP --export-all
and SHASTA_HIP_LIB=tools/probes/_bin/libshasta_pstamp.so.  Every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz)
around itself; clock = d(s_memtime) / d(s_memrealtime) * 100 MHz, median over the 4096 workgroups of the last launch, after 2 s of
back-to-back launches of (a) the whole forward, (b) the pair stage alone, and (c) one launch after 300 ms of idle.
usage: pair_clock.py [B] [arithmetic]   (arithmetic "pieces": pair_mfma4_kernel, diagnostic build of pair.hip instead)"""
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np
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
if len(sys.argv) > 2:
    m.arithmetic = sys.argv[2]
lib = hip.load()
dbg = C.CDLL(os.environ["SHASTA_HIP_LIB"])
w = m._weights()
m._ensure_packed(w, dev)
N, F, T = 500, 256, 502
g = torch.Generator(device=dev).manual_seed(1)
bev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
pbev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))


def boxes():
    b = torch.zeros(B, N, 11, device=dev)
    b[..., 0:2] = torch.rand(B, N, 2, device=dev, generator=g) * 100 - 50
    b[..., 2] = torch.randn(B, N, device=dev, generator=g)
    b[..., 3:6] = torch.rand(B, N, 3, device=dev, generator=g) * 4 + 0.5
    b[..., 6] = (torch.rand(B, N, device=dev, generator=g) * 2 - 1) * 3.14159265
    b[..., 7:9] = torch.randn(B, N, 2, device=dev, generator=g)
    b[..., 9] = 0.5
    return b


det, prev = boxes(), boxes()
feat = torch.rand(B, T, F, device=dev)
pfeat = torch.rand(B, T, F, device=dev)
dt = torch.rand(B, T, 8, device=dev) * 4 + 0.5
pt = torch.rand(B, T, 8, device=dev) * 4 + 0.5
res = torch.empty(B, T, 504, device=dev)
wsb = lib.shasta_forward_workspace_bytes(B, N, 7, F)
ws = torch.empty(wsb // 4 + 1, device=dev)


def stage():
    hip.check(lib.shasta_pair_residual_f32(C.byref(w), hip.ptr(m._packed), B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(dt),
                                           hip.ptr(pt), hip.ptr(res), 504, hip.ptr(ws), wsb, hip.stream_ptr()), "pair")


def forward():
    with torch.no_grad():
        m.affinity_from_bev(bev, pbev, det.clone(), prev)


def stamps(label, ms):
    buf = (C.c_ulonglong * (4096 * 8 * 4))()
    assert dbg.shasta_debug_pair_stamp(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8, 4)
    a = a[a[:, 0, 2] > 0]
    cyc, t0, t1 = a[:, :, 0].astype(np.float64), a[:, :, 1].astype(np.float64) / 100, a[:, :, 2].astype(np.float64) / 100  # us
    clk = (cyc / (t1 - t0) * 1e-3)[:, 0]  # GHz
    w0, w1 = t0.min(1), t1.max(1)           # workgroup = first wave start .. last wave end
    dur = w1 - w0
    hw = a[:, 0, 3]
    cu = ((hw >> np.uint64(32)) & np.uint64(15)) * np.uint64(1 << 16) + (hw & np.uint64(0xff00))  # (XCC, SE, SH, CU)
    print("    waves of the first workgroup: " + " ".join("w%d simd %d slot %d" % (w, (int(a[0, w, 3]) >> 4) & 3, int(a[0, w, 3]) & 15) for w in range(8)))
    ids, counts = np.unique(cu, return_counts=True)
    gaps = []
    for c in ids:
        sel = np.argsort(w0[cu == c])
        s0, s1 = w0[cu == c][sel], w1[cu == c][sel]
        gaps += list(s0[1:] - s1[:-1])
    gaps = np.array(gaps)
    span = w1.max() - w0.min()
    print("%-38s %.3f ms per call | clock median %.3f GHz (p10 %.3f p90 %.3f) | wave %.0f k cycles | workgroup %.1f us (p90 %.1f); its waves "
          "start within %.1f us and end within %.1f us (medians; p90 %.1f) | first start to last end %.3f ms on %d CUs, %d .. %d workgroups "
          "per CU, busy %.2f | gap last wave end -> next workgroup's first wave start on a CU: median %.1f p90 %.1f max %.1f us" %
          (label, ms, np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90), np.median(cyc) / 1e3, np.median(dur), np.percentile(dur, 90),
           np.median(t0.max(1) - t0.min(1)), np.median(t1.max(1) - t1.min(1)), np.percentile(t1.max(1) - t1.min(1), 90), span / 1e3, len(ids),
           counts.min(), counts.max(), dur.sum() / (span * len(ids)), np.median(gaps), np.percentile(gaps, 90), gaps.max()))
    order = np.argsort(np.median(t1, 0))
    print("    median end of wave w after the workgroup's first wave end (us): " +
          " ".join("w%d %.1f" % (w, np.median(t1[:, w] - t1.min(1))) for w in range(8)))


def loop(fn, seconds):
    fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.time()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.time() - t0 < seconds:
        a.record()
        for _ in range(10):
            fn()
        b.record()
        b.synchronize()
        n += 10
    return a.elapsed_time(b) / 10


stamps("whole forward, back to back for 2 s", loop(forward, 2.0))
stamps("pair stage alone, back to back for 2 s", loop(stage, 2.0))
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    time.sleep(0.3)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    stage()
    b.record()
    b.synchronize()
    ts.append(a.elapsed_time(b))
stamps("pair stage once after 300 ms idle", statistics.median(ts))
stamps("whole forward again", loop(forward, 2.0))
