#!/bin/bash
# round 4, pass f: K0 tests (every arithmetic, bank, range) + the bench line with the new extras
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4f}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fuzz.py tests/test_abi.py -m gpu -q -x --tb=short -k "${2:-shared_conv or abi}" > $O/pytest.log 2>&1
tail -6 $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value %.0f ms/step %.3f" % (d["value"], d["ms_per_step"]))
ex=d.get("extra",{})
print(json.dumps(ex.get("shared_conv"), indent=1)[:3000])
print(json.dumps(ex.get("car_90_320_3",{}).get("from_neck_b1")))
PY
