#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "pair or forward or arithmetic or golden" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
for round in 1 2; do for lib in shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_prev.so; do
  SHASTA_HIP_LIB=$R/$lib python3 - <<PY 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
import bench, sys, types, torch
args = types.SimpleNamespace(batch=512, no_precut=False)
b = bench.Bench(args, torch.device("cuda", 0), 0, 1, None)
r0 = r1 = dict(value=0, pair_ms=0)
r2 = b.measure(bench.CAR, 512, 200, 20, "f16x2")
r3 = b.measure(bench.CAR, 1, 200, 5, "f16x2")
print("%-24s N=500 b512: %.0f fp/s pair %.3f ms | pieces: %.0f fp/s pair %.3f ms | car b512: %.0f fp/s pair %.3f ms | car b1 %.4f ms pair %.4f" % ("$lib".split("/")[-1], r0["value"], r0["pair_ms"], r1["value"], r1["pair_ms"], r2["value"], r2["pair_ms"], r3["ms_per_step"], r3["pair_ms"]))
PY
done; done
