#!/bin/bash
# GPU-box helper: the training part of tools/measure_round.sh alone (kernel stats of the step at N=500 x 8 and the car configuration x 64)
# into gpurun_out/<tag>/ - for refreshing profiles/<round>_kernel_stats_train_* and <round>_train_*.txt after a change to the backward
# only (tools/refresh_train_profiles.py merges them).  usage: bash tools/gpu_train_refresh.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r6t}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_n500 -o d -- python3 $R/tools/time_train.py --max-obj 500 --feats 7 --points 4 --batch 8 --steps 8 --in-backward > $O/train_n500.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_n90 -o d -- python3 $R/tools/time_train.py --max-obj 90 --feats 3 --points 5 --batch 64 --steps 8 --in-backward > $O/train_n90.log 2>&1
grep -h "ms/step" $O/train_n500.log $O/train_n90.log | cut -c1-200
(cd $R && python3 tools/time_gather_bwd.py 2>&1 | grep us | tee $O/gather_bwd.log)
(cd $R && python3 tools/probes/train_determinism.py 2>&1 | tail -3 | tee $O/train_determinism.log)
(cd $R && python3 tools/train_soak_conv.py 2>&1 | tail -3 | tee $O/train_soak_conv.log)
