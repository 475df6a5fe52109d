"""Experiment: two independent forward pipelines on two halves of the chip (CU-masked HIP streams), half a step out of
phase, against one pipeline on the whole chip.  The step is power-limited with a very uneven power profile (bf16-piece weight
stream vs pair kernel); running the two phases side by side flattens it.
usage: python tools/two_pipelines.py [--batch 128] [--steps 200] [--mode masked|plain|single] [--offset-ms 2.0]"""
import argparse
import copy
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--mode", default="masked", choices=["masked", "plain", "single", "interleaved"])
ap.add_argument("--offset-ms", type=float, default=2.0)
ap.add_argument("--pipes", type=int, default=2, help="number of pipelines in plain mode")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = 500
with torch.device(dev):
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                         bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                         max_obj=N, num_feats=7, num_point=4)).eval()
m2 = copy.copy(m)
m2._bufs = {}


def inputs(seed, B):
    g = torch.Generator(device=dev).manual_seed(seed)
    bev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
    pbev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
    det0 = torch.zeros(B, N, 11, device=dev)
    det0[..., :2] = torch.rand(B, N, 2, device=dev, generator=g) * 100 - 50
    det0[..., 3:6] = torch.rand(B, N, 3, device=dev, generator=g) * 4 + 0.5
    det0[..., 6] = torch.rand(B, N, device=dev, generator=g) * 6.28 - 3.14
    det0[..., 9] = 0.5
    return bev, pbev, det0, det0.roll(1, 1).contiguous(), det0.clone()


def masked_stream(words):
    hipr = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hipr.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
    assert rc == 0, "hipExtStreamCreateWithCUMask rc=%d" % rc
    return torch.cuda.ExternalStream(st.value)


B = a.batch
if a.mode == "single":
    pipes = [(m, inputs(1, B), torch.cuda.Stream())]
elif a.mode == "plain":
    pipes = []
    for i in range(a.pipes):
        mi = m if i == 0 else copy.copy(m)
        if i:
            mi._bufs = {}
        pipes.append((mi, inputs(1 + i, B), torch.cuda.Stream()))
elif a.mode == "interleaved":  # every other CU
    pipes = [(m, inputs(1, B), masked_stream([0x55555555] * 8)), (m2, inputs(2, B), masked_stream([0xAAAAAAAA] * 8))]
else:  # first / second half of the CU bit range
    pipes = [(m, inputs(1, B), masked_stream([0xFFFFFFFF] * 4 + [0] * 4)), (m2, inputs(2, B), masked_stream([0] * 4 + [0xFFFFFFFF] * 4))]


def step(p):
    mod, (bev, pbev, det0, prev, det), st = p
    with torch.cuda.stream(st):
        det.copy_(det0, non_blocking=True)
        return mod.affinity_from_bev(bev, pbev, det, prev)


with torch.no_grad():
    for p in pipes:
        for _ in range(3):
            step(p)
    torch.cuda.synchronize()
    if len(pipes) >= 2 and a.offset_ms > 0:
        for i in range(1, len(pipes)):
            with torch.cuda.stream(pipes[i][2]):
                torch.cuda._sleep(int(a.offset_ms * i / (len(pipes) - 1 + 1) * 1e-3 * 100e6))  # ~100 MHz counter
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for p in pipes:
            out = step(p)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
print(json.dumps(dict(mode=a.mode, batch=B, pipelines=len(pipes), steps=a.steps, ms_per_round=el / a.steps * 1e3,
                      frame_pairs_per_s=round(len(pipes) * B * a.steps / el, 1), finite=bool(torch.isfinite(out[0]).all()))))
