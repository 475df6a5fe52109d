#!/bin/bash
# PMC passes that resolve the 8.4 GB vs 16.4 GB question for the weight stream at 512 frame-pairs per launch (four weight passes):
# fabric-side read requests by size and DRAM-bound requests, L2 hits / misses, for B=128 (one pass) and B=512 (four passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 128 512; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $O/pmc_rd_b$b -o p -- python3 $R/bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_rd_b$b.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/pmc_rd2_b$b -o p -- python3 $R/bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_rd2_b$b.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_hit_b$b -o p -- python3 $R/bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_hit_b$b.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_b$b -o p -- python3 $R/bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_fetch_b$b.log 2>&1
done
ls -R $O | head -50
