#!/bin/bash
# round 4: kernel-only times (rocprofv3 kernel stats) of the pair stage at F = 320: car tables x 512 frame-pairs, N = 500 x 256
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4pwt}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/pair_check.py --max-obj 150 --points 5 --feats 3 --batch 2 2>&1 | grep "^{" | head -2 | cut -c1-150
for cfg in "90 512" "500 256"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o d -- python3 $R/tools/pair_time.py --max-obj $1 --batch $2 --iters 20 > $O/time_$1.log 2>&1
  grep "^{" $O/time_$1.log
  f=$(ls $O/prof_$1/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "pair_f16w_kernel|pair_mfma4_kernel|embed_rows|row_prep|gemm" $f | cut -c1-200 | sed -E 's/\(.*\)//' | awk -F, '{printf "%-60s calls %s avg_ns %s\n", substr($1,1,60), $2, $4}'
done
