"""Times the per-pair MLP kernels of the training backward (csrc/pair_bwd.hip) alone, events on the stream.
usage: python tools/time_pair_mlp.py [--max-obj 500] [--feat 256] [--batch 8] [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shasta_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--max-obj", type=int, default=500)
ap.add_argument("--feat", type=int, default=256)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
lib = hip.load()
dev = torch.device("cuda:0")
F, B, T = a.feat, a.batch, a.max_obj + 2
WIDTHS = {0: (F // 8, F // 16, F // 32, 1), 1: (32, 8, 1), 2: (32 + F // 8, 8 + F // 32, 3)}
out = {}
for kind, widths in WIDTHS.items():
    torch.manual_seed(kind)
    UP, UC = torch.randn(B * T, widths[0], device=dev), torch.randn(B * T, widths[0], device=dev)
    layers = [(torch.randn(widths[i + 1], widths[i], device=dev) / widths[i] ** 0.5, torch.randn(widths[i + 1], device=dev)) for i in range(len(widths) - 1)]
    flat = [t for wb in layers for t in wb] + [None] * (6 - 2 * len(layers))
    wt = (C.c_void_p * 6)(*[None if t is None else t.data_ptr() for t in flat])
    res = torch.empty(B * T * T, widths[-1], device=dev)
    gout = torch.randn(B * T * T, widths[-1], device=dev)
    nb = lib.shasta_pair_mlp_workspace_bytes(kind, F, B, T, T)
    ws = torch.empty((nb + 3) // 4, device=dev)
    gUP, gUC = torch.empty_like(UP), torch.empty_like(UC)
    img = torch.empty(lib.shasta_pair_mlp_grad_floats(kind, F), device=dev)

    def fwd():
        hip.check(lib.shasta_pair_mlp_forward_f32(kind, F, hip.ptr(UP), hip.ptr(UC), wt, B, T, T, hip.ptr(res), hip.stream_ptr()), "fwd")

    def bwd():
        hip.check(lib.shasta_pair_mlp_backward_f32(kind, F, hip.ptr(UP), hip.ptr(UC), wt, hip.ptr(gout), B, T, T, hip.ptr(gUP), hip.ptr(gUC), hip.ptr(img),
                                                   hip.ptr(ws), nb, hip.stream_ptr()), "bwd")

    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out["%s%d" % (name, kind)] = e0.elapsed_time(e1) / a.iters
print(os.environ.get("SHASTA_HIP_LIB", "default"), " ".join("%s %.3f" % kv for kv in out.items()), "(ms; kinds 0 fuse_shape, 1 fuse_det, 2 res_coeff)")
