"""Experiment: inside ONE step, run the aug_shape stage (the power-capped 4.1 GB weight stream, matrix pipes) on a side stream
while the pair stage (bound by vector issue, 170 - 270 W below the cap when alone) runs on the main stream.  Only rows N, N + 1 of
the feature tables - 2 of 502 - come out of aug_shape, so all but 0.8 % of the pair work does not depend on it.  This probe
measures what the overlap is worth BEFORE the split is built: the pair stage here simply reads whatever the anchor rows hold
(results are not checked; timing only).  CU masks restrict the side stream to a share of the chip.
usage: python tools/overlap_probe.py [--batch 512] [--steps 30] [--rounds 2]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402
from shasta_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, NF, NP, F = 500, 7, 4, 256
T = N + 2
with torch.device(dev):
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                         bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                         max_obj=N, num_feats=NF, num_point=NP)).eval()
lib = hip.load()
B = a.batch
g = torch.Generator(device=dev).manual_seed(1)
bev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
pbev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
det0 = torch.zeros(B, N, 11, device=dev)
det0[..., :2] = torch.rand(B, N, 2, device=dev, generator=g) * 100 - 50
det0[..., 3:6] = torch.rand(B, N, 3, device=dev, generator=g) * 4 + 0.5
det0[..., 6] = torch.rand(B, N, device=dev, generator=g) * 6.28 - 3.14
det0[..., 9] = 0.5
prev = det0.roll(1, 1).contiguous()
det = det0.clone()
w = m._weights()
m._ensure_packed(w, dev)
m._ensure_aux(w, B, dev)
feat = torch.zeros(B, T, F, device=dev)
pfeat = torch.zeros(B, T, F, device=dev)
dtab = torch.zeros(B, T, 8, device=dev)
ptab = torch.zeros(B, T, 8, device=dev)
Dp = (T + 3) // 4 * 4
res = torch.zeros(B, T, Dp, device=dev)
m1 = torch.empty(B, N, T, device=dev)
m2 = torch.empty(B, T, N, device=dev)
wsb = lib.shasta_forward_workspace_bytes(B, N, NF, F)
ws_a = torch.empty(wsb // 4 + 1, device=dev)   # aug_shape stage (side stream)
ws_b = torch.empty(wsb // 4 + 1, device=dev)   # everything else


def masked_stream(words):
    hipr = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hipr.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
    assert rc == 0, "hipExtStreamCreateWithCUMask rc=%d" % rc
    return torch.cuda.ExternalStream(st.value)


def sp(s):
    return C.c_void_p(s.cuda_stream)


main = torch.cuda.Stream()


def step(side):
    with torch.cuda.stream(main):
        det.copy_(det0, non_blocking=True)
        m.bev_extractor.gather_boxes(bev, det, NP, feat)
        m.bev_extractor.gather_boxes(pbev, prev, NP, pfeat)
        s2 = main if side is None else side
        if side is not None:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
        hip.check(lib.shasta_anchor_shape_f32(C.byref(w), B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(ws_a), wsb, sp(s2)), "anchor_shape")
        hip.check(lib.shasta_anchor_boxes_f32(C.byref(w), B, hip.ptr(det), hip.ptr(prev), 11, hip.ptr(dtab), hip.ptr(ptab), hip.ptr(ws_b), wsb, sp(main)),
                  "anchor_boxes")
        hip.check(lib.shasta_pair_residual_f32(C.byref(w), hip.ptr(m._packed), B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(dtab), hip.ptr(ptab),
                                               hip.ptr(res), Dp, hip.ptr(ws_b), wsb, sp(main)), "pair")
        if side is not None:
            ev2 = torch.cuda.Event()
            ev2.record(side)
            main.wait_event(ev2)
        hip.check(lib.shasta_aff_softmax_f32(C.byref(w), hip.ptr(m._packed), B, hip.ptr(res), Dp, hip.ptr(m1), hip.ptr(m2), None, hip.ptr(ws_b), wsb,
                                             sp(main)), "aff")


def byte_mask(k):      # k of every 8 consecutive CU bits
    return [int.from_bytes(bytes([(1 << k) - 1] * 4), "little")] * 8


def word_mask(k):      # the first k of the 8 mask words
    return [0xFFFFFFFF] * k + [0] * (8 - k)


modes = [("sequential", None), ("side stream, no mask", torch.cuda.Stream())]
for k in (2, 3, 4):
    modes.append(("side stream, %d of every 8 CU bits" % k, masked_stream(byte_mask(k))))
for k in (2, 3, 4):
    modes.append(("side stream, first %d of 8 mask words" % k, masked_stream(word_mask(k))))
with torch.no_grad():
    for name, side in modes:
        for _ in range(3):
            step(side)
    torch.cuda.synchronize()
    for rnd in range(a.rounds):
        for name, side in modes:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step(side)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            print(json.dumps(dict(mode=name, round=rnd, batch=B, ms_per_step=round(el / a.steps * 1e3, 3),
                                  frame_pairs_per_s=round(B * a.steps / el, 1))), flush=True)
