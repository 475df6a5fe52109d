#!/bin/bash
# counters of the weight-stream kernel and of the aff kernel at 1024 frame-pairs: bash tools/gpu_pmc_l1.sh <subdir of gpurun_out>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_l1}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="python3 $R/bench.py --batch 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o p -- $A > $O/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $A > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $A > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/wait -o p -- $A > $O/wait.log 2>&1
for k in anchor_l1 aff_frame; do for d in mfma fetch write wait; do python3 $R/tools/pmc_table.py $O/$d $k; done; done
