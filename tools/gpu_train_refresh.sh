#!/bin/bash
# Re-collects the training files of a measurement pass (kernel stats of a training step at N=500 B=8 and at the car configuration x 64, the
# per-pair MLP kernels and the low-rank Adam pass alone) and the driver line into gpurun_out/r5m3.  usage (GPU box): bash tools/gpu_train_refresh.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5m3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_train_n500 $O/prof_train_n90
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_n500 -o d -- python3 $R/tools/time_train.py --max-obj 500 --feats 7 --points 4 --batch 8 --steps 8 --in-backward > $O/train_n500.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_n90 -o d -- python3 $R/tools/time_train.py --max-obj 90 --feats 3 --points 5 --batch 64 --steps 8 --in-backward > $O/train_n90.log 2>&1
(cd $R && python3 tools/time_pair_mlp.py > $O/pair_mlp.log 2>&1; python3 tools/time_adam_lowrank.py >> $O/pair_mlp.log 2>&1; python3 tools/time_adam_lowrank.py --dx >> $O/pair_mlp.log 2>&1; python3 tools/time_adam_lowrank.py --rank 64 >> $O/pair_mlp.log 2>&1)
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
grep -h "ms/step" $O/train_n500.log $O/train_n90.log; cat $O/pair_mlp.log | grep -v amdgpu
