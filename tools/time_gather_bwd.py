"""GPU-box helper: shasta_bev_gather_bwd_f32 alone (HIP events), the two shapes of the training profiles.
usage: [SHASTA_HIP_LIB=variant.so] python tools/time_gather_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import shasta_oracle as O  # noqa: E402
from shasta_amd import hip  # noqa: E402

lib = hip.load()
dev = torch.device("cuda:0")
for B, N, npnt in ((8, 500, 4), (64, 90, 5), (16, 90, 5), (1, 500, 4)):
    C, H, W = 64, 180, 180
    g = torch.Generator().manual_seed(0)
    boxes = O.synth_boxes(g, B, N, None)[:, :, :7].contiguous().to(dev)
    F = npnt * C
    dfeat = torch.randn(B, N, F, generator=g).to(dev)
    dbev = torch.zeros(B, H, W, C, device=dev)

    def run():
        dbev.zero_()
        hip.check(lib.shasta_bev_gather_bwd_f32(hip.ptr(dfeat), B, H, W, C, hip.ptr(boxes), N, 7, N * 7, npnt, -54.0, -54.0, 0.075, 0.075, 8.0, F, N * F,
                                                hip.ptr(dbev), hip.stream_ptr()), "shasta_bev_gather_bwd_f32")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        dbev.zero_()
        e0.record()
        hip.check(lib.shasta_bev_gather_bwd_f32(hip.ptr(dfeat), B, H, W, C, hip.ptr(boxes), N, 7, N * 7, npnt, -54.0, -54.0, 0.075, 0.075, 8.0, F, N * F,
                                                hip.ptr(dbev), hip.stream_ptr()), "shasta_bev_gather_bwd_f32")
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("B=%d N=%d points=%d: %.1f us (median of 20, events around the launch)" % (B, N, npnt, ts[10] * 1e3), flush=True)
