"""GPU-box helper: the shipped car configuration (max_obj 90, F = 320, nf = 3) at batch 1 / 8, for rocprofv3 --kernel-trace --stats.
usage: car_b1.py [B] [steps]"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
args = types.SimpleNamespace(batch=max(B, 8), no_precut=False)
bn = bench.Bench(args, torch.device("cuda", 0), 0, 1, None)
r = bn.measure(bench.CAR, B, steps, 10, "f16x2")
print("car B=%d: %.4f ms per step (%.0f fp/s), weight stream %.4f ms, pair %.4f ms" % (B, r["ms_per_step"], r["value"], r["l1_ms"], r["pair_ms"]))
r = bn.measure(bench.CAR, B, steps, 10, "f16x2", graph=True)
print("car B=%d, hipGraph replay: %.4f ms per step (%.0f fp/s)" % (B, r["ms_per_step"], r["value"]))
if os.environ.get("CAR_B1_ONE_PASS"):
    from shasta_amd import hip  # noqa: E402
    bn.model(bench.CAR).extra_options = hip.OPT_ONE_PASS_AFF
    bn.model(bench.CAR).invalidate_weights_cache()
    r = bn.measure(bench.CAR, B, steps, 10, "f16x2", graph=True)
    print("car B=%d, one-pass aff, hipGraph replay: %.4f ms per step (%.0f fp/s)" % (B, r["ms_per_step"], r["value"]))
