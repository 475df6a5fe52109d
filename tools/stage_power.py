"""GPU-box helper: each stage of the affinity forward looped ALONE for about `seconds` of GPU time, with the device's energy
accumulator read around the loop: ms / call, joules / call and average power of the stage when it has the chip to itself, next to
the same for the whole forward.  usage: python tools/stage_power.py [B] [seconds]"""
import ctypes as C
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.getcwd())
import shasta_amd  # noqa: E402
from bench import EnergyCounter  # noqa: E402
from shasta_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, NF, NP, CH, HW = 500, 7, 4, 64, 180
with torch.device(dev):
    m = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
        bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
        max_obj=N, num_feats=NF, num_point=NP)).eval()
lib = hip.load()
w = m._weights()
m._ensure_packed(w, dev)
m._ensure_aux(w, B, dev)
F, T = CH * NP, N + 2
g = torch.Generator(device=dev).manual_seed(1)
feat = torch.rand(B, T, F, device=dev, generator=g)
pfeat = torch.rand(B, T, F, device=dev, generator=g)
dt = torch.rand(B, T, 8, device=dev, generator=g) * 4 + 0.5
pt = torch.rand(B, T, 8, device=dev, generator=g) * 4 + 0.5
res = torch.randn(B, T, 504, device=dev, generator=g)
m1 = torch.empty(B, N, T, device=dev)
m2 = torch.empty(B, T, N, device=dev)
wsb = lib.shasta_forward_workspace_bytes(B, N, NF, F)
ws = torch.empty(wsb // 4 + 1, device=dev)
bev = torch.relu(torch.randn(B, HW, HW, CH, device=dev, generator=g))
pbev = torch.relu(torch.randn(B, HW, HW, CH, device=dev, generator=g))
det0 = torch.zeros(B, N, 11, device=dev)
det0[..., 0:2] = torch.rand(B, N, 2, device=dev, generator=g) * 100 - 50
det0[..., 3:6] = torch.rand(B, N, 3, device=dev, generator=g) * 4 + 0.5
prev = det0.clone()
det = det0.clone()
st = hip.stream_ptr()
wp, pk = C.byref(w), hip.ptr(m._packed)


def anchors():
    hip.check(lib.shasta_anchor_shape_f32(wp, B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(ws), wsb, st), "anchor_shape")


def pair():
    hip.check(lib.shasta_pair_residual_f32(wp, pk, B, hip.ptr(feat), hip.ptr(pfeat), hip.ptr(dt), hip.ptr(pt), hip.ptr(res), 504,
                                           hip.ptr(ws), wsb, st), "pair")


def aff():
    hip.check(lib.shasta_aff_softmax_f32(wp, pk, B, hip.ptr(res), 504, hip.ptr(m1), hip.ptr(m2), None, hip.ptr(ws), wsb, st), "aff")


def forward():
    det.copy_(det0)
    with torch.no_grad():
        m.affinity_from_bev(bev, pbev, det, prev)


energy = EnergyCounter(0)


class Freq(C.Structure):  # rsmi_frequencies_t
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]


class Sampler(threading.Thread):
    """sclk (rsmi_dev_gpu_clk_freq_get, RSMI_CLK_TYPE_SYS) sampled while a loop runs"""

    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.mhz = False, []

    def run(self):
        f = Freq()
        while not self.stop and energy.lib is not None:
            if energy.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(0), C.c_int(0), C.byref(f)) == 0 and f.current < 33:
                self.mhz.append(f.frequency[f.current] / 1e6)
            time.sleep(0.02)


out = {}
for name, fn in (("forward", forward), ("anchor_shape", anchors), ("pair_residual", pair), ("aff_softmax", aff), ("forward_again", forward)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    iters = max(5, int(seconds / max(time.perf_counter() - t0, 1e-5)))
    smp = Sampler()
    smp.start()
    e0 = energy.joules()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    e1 = energy.joules()
    smp.stop = True
    smp.join()
    mhz = sorted(smp.mhz[len(smp.mhz) // 4:]) or [float("nan")]
    j = (e1 - e0) if e0 is not None and e1 is not None else float("nan")
    out[name] = {"ms": el / iters * 1e3, "joules": j / iters, "watts": j / el, "iters": iters, "sclk_mhz_median": mhz[len(mhz) // 2],
                 "sclk_mhz_min": mhz[0], "sclk_mhz_max": mhz[-1]}
    print("%-14s %8.3f ms / call  %7.3f J / call  %6.0f W  sclk median %.0f MHz (%.0f - %.0f, %d samples)  (%d calls)"
          % (name, el / iters * 1e3, j / iters, j / el, mhz[len(mhz) // 2], mhz[0], mhz[-1], len(mhz), iters), flush=True)
print(json.dumps({"B": B, "stages": out}))
