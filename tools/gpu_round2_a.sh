#!/bin/bash
# round 2, first GPU pass: full GPU test suite, a bench line, kernel stats, counter list
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q -rA 2>&1 | tail -150 > $O/pytest.log
tail -5 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o d -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_default.log 2>&1
rocprofv3 --list-avail > $O/counters.txt 2>&1
ls $O $O/prof_default | head -30
