#!/bin/bash
# round 4, pass d: K0 fp16 form after a kernel change - accuracy, timing, the conv parity tests, kernel durations
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4d}
mkdir -p $O
cd $R
python tools/conv_check.py --batches 1,8 > $O/base.jsonl 2> $O/base.err
grep -E '"time"|accuracy|equal' $O/base.jsonl | grep -v miopen | cut -c1-200
tail -3 $O/base.err
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_fuzz.py -m gpu -q -x --tb=short -k "shared_conv" > $O/pytest_conv.log 2>&1
tail -3 $O/pytest_conv.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o p -- python3 $R/tools/conv_only.py --batch 8 --iters 4 > $O/p1.log 2>&1
python3 $R/tools/pmc_table.py $O/p1 conv_f16
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p2 -o p -- python3 $R/tools/conv_only.py --batch 8 --iters 4 > $O/p2.log 2>&1
python3 $R/tools/pmc_table.py $O/p2 conv_f16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 $R/tools/conv_only.py --batch 8 --iters 10 > $O/prof.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/d_kernel_stats.csv")))
for r in rows[:4]:
    print("%-70s calls %4s avg_us %9.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
