#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for b in 512 768 1024 1536 2048 512; do
python bench.py --no-cpu-baseline --no-extras --steps 16 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=[d['roofline'],d['roofline_second']]
p=[x for x in r if x['kernel'].startswith('pair')][0]; l=[x for x in r if not x['kernel'].startswith('pair')][0]
print('B=%d %.0f fp/s step %.3f ms pair %.3f (%.3f) L1 %.3f (%.3f) %.2f mJ/fp %.0f W' % (d['config']['frame_pairs_per_step_per_gpu'], d['value'], d['ms_per_step'], p['avg_launch_ms'], p['frac'], l['avg_launch_ms'], l['frac'], d['energy']['millijoules_per_frame_pair'], d['energy']['avg_power_w']))"
done
