#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-l1modes}
mkdir -p $O
cd $R
for m in pieces f32 f16x2; do
  python tools/l1_split_check.py --max-obj 500 --batch 64 128 256 512 --steps 10 --arithmetic $m > $O/check_$m.json 2> $O/check_$m.err
  cat $O/check_$m.json
done
for m in pieces f16x2; do
  python bench.py --no-cpu-baseline --steps 30 --arithmetic $m > $O/bench_$m.json 2> $O/bench_$m.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$m.json").read().strip().splitlines()[-1])
r=[d["roofline"],d["roofline_second"]]
print("$m: %.0f fp/s step %.3f ms selfcheck %.1e | " % (d["value"], d["ms_per_step"], d["selfcheck_max_abs"]) + " | ".join("%s %.3f ms" % (x["kernel"][:12], x["avg_launch_ms"]) for x in r))
PY
done
