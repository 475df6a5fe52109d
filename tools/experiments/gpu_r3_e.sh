#!/bin/bash
# round 3, fifth GPU pass: pre-cut fp16 stream for small batches: tests + batch sweep with / without
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
python -m pytest tests/test_hip_parity.py tests/test_fuzz.py tests/test_pipeline.py tests/test_training.py -m gpu -q --tb=short -rf -x > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for b in 2 8 16 32 48 64 128; do
  for v in "" "--no-precut"; do
    python bench.py --no-cpu-baseline --no-extras --steps 60 --batch $b $v > $O/b${b}_${v#--}.json 2> $O/b${b}_${v#--}.err
    python - <<PY
import json
d=json.loads(open("$O/b${b}_${v#--}.json").read().strip().splitlines()[-1])
r=[d["roofline"],d["roofline_second"]]
p=[x for x in r if x["kernel"].startswith("pair")][0]; l=[x for x in r if not x["kernel"].startswith("pair")][0]
print("B=%-4d %-12s %.0f fp/s  step %.3f ms  pair %.3f ms  L1 %.3f ms (%s, frac %.3f)  selfcheck %.1e" % ($b, "${v:-precut}", d["value"], d["ms_per_step"], p["avg_launch_ms"], l["avg_launch_ms"], l["kernel"][:34], l["frac"], d["selfcheck_max_abs"]))
PY
  done
done
