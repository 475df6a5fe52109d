"""Timings of the widened rows on the GPU box: rotated IoU matrix, rotated NMS, batched decode decisions,
batched tracker assignment.  Device time by HIP events around the C-ABI calls (inputs resident), 20 repetitions each.
usage: python tools/time_rows.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shasta_amd import hip, nms  # noqa: E402
from shasta_amd.pub_tracker import center_greedy_device  # noqa: E402

dev = torch.device("cuda:0")
lib = hip.load()
rng = np.random.default_rng(0)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def boxes7(n, spread):
    b = np.zeros((n, 7))
    b[:, :2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.normal(0, 0.5, n)
    b[:, 3] = rng.uniform(-3.14, 3.14, n)
    b[:, 4:7] = rng.uniform(0.5, 5, (n, 3))
    return b


# rotated 3-D IoU / GIoU distance matrix (float64), 500 x 500
D, T = (torch.from_numpy(boxes7(500, 40)).to(dev) for _ in range(2))
out = torch.empty(500, 500, dtype=torch.float64, device=dev)
for mode, name in ((0, "iou3d"), (1, "giou3d")):
    us = timed(lambda: hip.check(lib.shasta_iou3d_distance_f64(hip.ptr(D), 500, hip.ptr(T), 500, 7, mode, hip.ptr(out), hip.stream_ptr()), "iou"))
    print("%-34s %8.1f us  (%.1f M pairs/s)" % (name + " distance matrix 500 x 500", us, 0.25 / us * 1e6))

# rotated BEV NMS
for n, spread in ((1000, 30), (4096, 60), (16000, 120)):
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 3:6] = rng.uniform(0.5, 5, (n, 3))
    b[:, 6] = rng.uniform(-3.14, 3.14, n)
    bs = torch.from_numpy(b).to(dev)
    keep = torch.empty(n, dtype=torch.int32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.shasta_nms_workspace_bytes(n)
    ws = torch.empty(wsb // 8 + 1, dtype=torch.float64, device=dev)
    us = timed(lambda: hip.check(lib.shasta_nms_rotated_f32(hip.ptr(bs), n, 0.2, hip.ptr(ws), wsb, hip.ptr(keep), hip.ptr(num), hip.stream_ptr()), "nms"), reps=5)
    print("%-34s %8.1f us  (kept %d)" % ("rotated NMS, %d boxes" % n, us, int(num.item())))

# batched decode decisions, 64 frames of 500 x 502
B, N = 64, 500
m1 = torch.softmax(torch.randn(B, N, N + 2, device=dev), dim=2)
m2 = torch.softmax(torch.randn(B, N + 2, N, device=dev), dim=1)
npv = torch.full((B,), 400, dtype=torch.int32, device=dev)
ncv = torch.full((B,), 420, dtype=torch.int32, device=dev)
pc, df = (torch.empty(B, N, dtype=torch.int32, device=dev) for _ in range(2))
ps, ds = (torch.empty(B, N, dtype=torch.float32, device=dev) for _ in range(2))
us = timed(lambda: hip.check(lib.shasta_decode_flags_f32(hip.ptr(m1), hip.ptr(m2), hip.ptr(npv), hip.ptr(ncv), B, N, hip.ptr(pc), hip.ptr(ps),
                                                          hip.ptr(df), hip.ptr(ds), hip.stream_ptr()), "decode"))
print("%-34s %8.1f us  (%.2f us per frame)" % ("decode decisions, 64 frames", us, us / B))

# tracker assignment: 150 scenes, 300 detections x 300 tracks each (device kernel only + end-to-end incl. host staging)
S, n, m = 150, 300, 300
probs = [(rng.uniform(-50, 50, (n, 2)).astype(np.float32), rng.uniform(-50, 50, (m, 2)).astype(np.float32), rng.integers(0, 7, n).astype(np.int32),
          rng.integers(0, 7, m).astype(np.int32), rng.choice(np.array([0.75, 2, 4], np.float32), n)) for _ in range(S)]
arrs = [torch.from_numpy(np.stack([p[k] for p in probs])).to(dev) for k in range(5)]
nn_ = torch.full((S,), n, dtype=torch.int32, device=dev)
mm_ = torch.full((S,), m, dtype=torch.int32, device=dev)
dist = torch.empty(S, n, m, dtype=torch.float64, device=dev)
match = torch.empty(S, n, dtype=torch.int32, device=dev)
us = timed(lambda: hip.check(lib.shasta_center_greedy_f32(hip.ptr(arrs[0]), hip.ptr(arrs[1]), hip.ptr(arrs[2]), hip.ptr(arrs[3]), hip.ptr(arrs[4]),
                                                           hip.ptr(nn_), hip.ptr(mm_), S, n, m, hip.ptr(dist), hip.ptr(match), None, None, hip.stream_ptr()), "greedy"), reps=5)
print("%-34s %8.1f us  (%.1f us per scene)" % ("tracker assign, 150 scenes 300x300", us, us / S))
import time  # noqa: E402
for want in (True, False):
    t0 = time.perf_counter()
    center_greedy_device(probs, want_dist=want)
    torch.cuda.synchronize()
    print("%-34s %8.1f us  end to end incl. host staging and D2H" % ("  center_greedy_device dist=%s" % want, (time.perf_counter() - t0) * 1e6))
