"""GPU-box helper: shasta_adam_lowrank_f32 alone on one (H, K) matrix (default: a first aug_shape layer at N = 500: 2000 x 128000).
usage: python tools/time_adam_lowrank.py [--rows 2000] [--cols 128000] [--rank 8] [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shasta_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2000)
ap.add_argument("--cols", type=int, default=128000)
ap.add_argument("--rank", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dx", action="store_true", help="shasta_adam_lowrank_dx_f32: Y += Gdx W in the same pass")
a = ap.parse_args()
lib = hip.load()
dev = torch.device("cuda:0")
H, K, R = a.rows, a.cols, a.rank
ps = [torch.randn(H, K, device=dev) for _ in range(2)]  # two matrices in turn: nothing of the next one is in a cache
ms = [torch.zeros(H, K, device=dev) for _ in range(2)]
vs = [torch.zeros(H, K, device=dev) for _ in range(2)]
G, X = torch.randn(R, H, device=dev), torch.randn(R, K, device=dev)


Y = torch.zeros(R, K, device=dev)
nb = lib.shasta_adam_lowrank_dx_workspace_bytes(H, K, R)
ws = torch.empty((nb + 3) // 4, device=dev)


def step(i, n):
    if a.dx:
        hip.check(lib.shasta_adam_lowrank_dx_f32(hip.ptr(ps[i]), hip.ptr(ms[i]), hip.ptr(vs[i]), H, K, hip.ptr(G), H, hip.ptr(X), K, R, hip.ptr(G), H, R, hip.ptr(Y), K, 1,
                                                 hip.ptr(ws), nb, 1e-4, 0.9, 0.999, 1e-8, 0.0, n, None, hip.stream_ptr()), "adam_lowrank_dx")
        return
    hip.check(lib.shasta_adam_lowrank_f32(hip.ptr(ps[i]), hip.ptr(ms[i]), hip.ptr(vs[i]), H, K, hip.ptr(G), H, hip.ptr(X), K, R, 1e-4, 0.9, 0.999, 1e-8, 0.0,
                                          n, None, hip.stream_ptr()), "adam_lowrank")


for n in range(1, 4):
    step(n & 1, n)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for n in range(a.iters):
    step(n & 1, 4 + n)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / a.iters
print("%s: adam_lowrank%s (%d x %d, R = %d) %.3f ms = %.2f TB/s over p, m, v in and out" % (os.environ.get("SHASTA_HIP_LIB", "default").split("/")[-1], "_dx" if a.dx else "", H, K, R, t, 24.0 * H * K / t / 1e9))
