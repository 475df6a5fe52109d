"""GPU-box helper: when and where the workgroups of the weight-stream kernel (anchor_l1_split_kernel) ran inside the real step.  Needs
   python tools/build_variant.py l1tl anchor_split.hip -DSHASTA_L1_TIMELINE --export-all
and SHASTA_HIP_LIB=tools/probes/_bin/libshasta_l1tl.so.  usage: l1_timeline.py [B]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
import types  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
args = types.SimpleNamespace(batch=B, no_precut=False)
bn = bench.Bench(args, torch.device("cuda", 0), 0, 1, None)
dbg = C.CDLL(os.environ["SHASTA_HIP_LIB"])
t0 = time.time()
while time.time() - t0 < 2.0:
    r = bn.measure(bench.HEADLINE, B, 20, 2, "f16x2")
print("step %.3f ms, weight stream %.3f ms, pair %.3f ms" % (r["ms_per_step"], r["l1_ms"], r["pair_ms"]))
buf = (C.c_ulonglong * (4096 * 4))()
assert dbg.shasta_debug_l1_timeline(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4)
a = a[a[:, 2] > 0]
cyc, s0, s1 = a[:, 0].astype(np.float64), a[:, 1].astype(np.float64) / 100, a[:, 2].astype(np.float64) / 100
dur = s1 - s0
clk = cyc / dur * 1e-3
hw = a[:, 3]
cu = ((hw >> np.uint64(32)) & np.uint64(15)) * np.uint64(1 << 16) + (hw & np.uint64(0xff00))
xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(int)
ids, counts = np.unique(cu, return_counts=True)
span = s1.max() - s0.min()
print("%d workgroups on %d CUs (%d .. %d per CU); first start to last end %.3f ms; workgroup %.1f us median (p10 %.1f p90 %.1f max %.1f); "
      "clock median %.3f GHz (p10 %.3f p90 %.3f); busy %.2f" %
      (len(a), len(ids), counts.min(), counts.max(), span / 1e3, np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.max(),
       np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90), dur.sum() / (span * 256)))
rel0, rel1 = s0 - s0.min(), s1 - s0.min()
order = np.argsort(rel0)
q = len(a) // 8
print("start times (us) of the workgroups in dispatch order, octiles: " + " ".join("%.0f" % rel0[order][min(i * q, len(a) - 1)] for i in range(9)))
print("end times (us), octiles of the sorted ends: " + " ".join("%.0f" % np.sort(rel1)[min(i * q, len(a) - 1)] for i in range(9)))
for x in range(8):
    sel = xcc == x
    if sel.any():
        print("  XCC %d: %3d workgroups, median %.1f us, last end %.0f us, clock %.3f GHz" % (x, sel.sum(), np.median(dur[sel]), rel1[sel].max(), np.median(clk[sel])))
