#!/bin/bash
# GPU-box helper: the K1 part of tools/measure_round.sh alone (kernel stats, the timing log, FETCH_SIZE / WRITE_SIZE passes) into
# gpurun_out/<tag>/ - for refreshing the voxeliser's committed summaries after a change to voxelize.hip only
# (tools/refresh_vox_profiles.py merges them into profiles/).  usage: bash tools/gpu_vox_refresh.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r6v}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
VOX_ITERS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_voxelize -o d -- python3 $R/tools/time_voxelize.py > $O/voxelize_prof.log 2>&1
(cd $R && python3 tools/time_voxelize.py > $O/voxelize.log 2>&1)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_voxfetch_b16 -o p -- python3 $R/tools/time_voxelize.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_voxwrite_b16 -o p -- python3 $R/tools/time_voxelize.py > /dev/null 2>&1
cat $O/voxelize.log
ls $O
