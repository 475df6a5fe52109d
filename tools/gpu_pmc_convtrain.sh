#!/bin/bash
# Kernel stats + counters of K0's train-mode kernels (tools/time_conv_train.py --only hand) at B frame pairs: bash tools/gpu_pmc_convtrain.sh <subdir> [B]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
B=${2:-8}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_convtrain_b$B -o d -- python3 $R/tools/time_conv_train.py --batch $B --only hand > $O/convtrain_b$B.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_convtrainmfma_b$B -o p -- python3 $R/tools/time_conv_train.py --batch $B --only hand --iters 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_convtrainlds_b$B -o p -- python3 $R/tools/time_conv_train.py --batch $B --only hand --iters 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_convtrainfetch_b$B -o p -- python3 $R/tools/time_conv_train.py --batch $B --only hand --iters 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_convtrainwrite_b$B -o p -- python3 $R/tools/time_conv_train.py --batch $B --only hand --iters 3 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_convtrainhit_b$B -o p -- python3 $R/tools/time_conv_train.py --batch $B --only hand --iters 3 > /dev/null 2>&1
for d in mfma lds fetch write hit; do python3 $R/tools/pmc_table.py $O/pmc_convtrain${d}_b$B conv_wgrad_kernel; done
