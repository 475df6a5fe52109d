"""GPU-box helper: why does `bench.py --steps 20 --warmup 5` read 2 % below `--steps 50`?  Times the energy-counter read and prints the
per-step durations of a 5 + 20 step run at the default batch (HIP events per step)."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = types.SimpleNamespace(batch=1024, no_precut=False)
dev = torch.device("cuda", 0)
bn = bench.Bench(args, dev, 0, 1, None)
en = bench.EnergyCounter(dev)
t0 = time.perf_counter()
for _ in range(20):
    en.joules()
print("energy counter read: %.3f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
for energy in (en, None, en, None):
    r = bn.measure(bench.HEADLINE, 1024, 20, 5, "f16x2", energy=energy)
    print("5 + 20 steps, energy read %s: %.3f ms per step (%.0f fp/s)" % ("on " if energy else "off", r["ms_per_step"], r["value"]))
r = bn.measure(bench.HEADLINE, 1024, 50, 5, "f16x2", energy=en)
print("5 + 50 steps: %.3f ms per step" % r["ms_per_step"])
# per-step durations
model = bn.model(bench.HEADLINE)
det0, prev = bn.boxes(bench.HEADLINE)
det = det0.clone()
torch.cuda.synchronize()
time.sleep(0.5)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(27)]
with torch.no_grad():
    evs[0].record()
    for i in range(26):
        det.copy_(det0)
        model.affinity_from_bev(bn.bev, bn.pbev, det, prev)
        evs[i + 1].record()
torch.cuda.synchronize()
print("26 consecutive steps after 0.5 s idle (ms): " + " ".join("%.2f" % evs[i].elapsed_time(evs[i + 1]) for i in range(26)))
