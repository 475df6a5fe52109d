#!/bin/bash
# round 4: K0 with two pixel blocks per wave (4 waves, -DC16_PB=2 variant build) against the default library; tracker + chain tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4pb}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_pub_tracker.py tests/test_pipeline.py -m gpu -q -x --tb=short > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for lib in shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_pb*.so; do
  echo "== $lib"
  SHASTA_HIP_LIB=$R/$lib timeout 300 python tools/conv_check.py --batches 1,8 --iters 10 2>/dev/null > $O/$(basename $lib .so).jsonl
  grep -E "accuracy" $O/$(basename $lib .so).jsonl | head -4 | cut -c20-200
  grep -E '"time"' $O/$(basename $lib .so).jsonl | grep f16x2 | cut -c1-200
done
