#!/bin/bash
# weight stream on v_mfma_f32_16x16x32_f16 (default build) against the 32x32x16 form (variant s32): parity tests, then A/B at 512 / 1024 / 64
cd "$(dirname "$0")/../.." && R=$PWD
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "precut or batch or operating_points or anchor" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
AB_STEPS=40 bash tools/gpu_ab.sh r3s shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_s32.so
for b in 512 64 32; do
for lib in shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_s32.so; do
  SHASTA_HIP_LIB=$R/$lib python bench.py --batch $b --no-cpu-baseline --no-extras --steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=[d['roofline'],d['roofline_second']]
l=[x for x in r if not x['kernel'].startswith('pair')][0]
print('B=$b %-40s %.0f fp/s step %.3f ms L1 %.3f ms frac %.3f  %s' % ('$lib'.split('/')[-1], d['value'], d['ms_per_step'], l['avg_launch_ms'], l['frac'], (d.get('energy') or {}).get('avg_power_w')))"
done; done
