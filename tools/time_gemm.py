"""Times the two NT GEMM entry points (f32 MFMA / bf16 pieces) on the shapes of the path.
usage: python tools/time_gemm.py [M N K ...]   (default: the row-embedding and aff layer shapes at 128 frame-pairs)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shasta_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
lib = hip.load()
v = [int(x) for x in sys.argv[1:]]
shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)] or [(64256, 128, 256), (64256, 128, 504), (64256, 504, 128), (64256, 64, 128)]
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    ref = (A.double() @ W.double().t() + b.double())
    for name in ("shasta_gemm_nt_f32", "shasta_gemm_nt_pieces_f32"):
        f = getattr(lib, name)
        for _ in range(3):
            hip.check(f(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(b), hip.ptr(C), N, M, N, K, 0, hip.stream_ptr()), name)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            f(hip.ptr(A), K, hip.ptr(W), K, hip.ptr(b), hip.ptr(C), N, M, N, K, 0, hip.stream_ptr())
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 20
        err = (C.double() - ref).abs().max().item()
        print("%-26s M=%d N=%d K=%d  %.1f us  %.1f TFLOP/s (fp32-equivalent)  max|err| %.2e (scale %.1f)"
              % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9, err, ref.abs().max().item()))
