#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for lib in shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_dbg_*.so; do
  echo "== $lib"
  SHASTA_HIP_LIB=$R/$lib python tools/conv_check.py --batches 1 --iters 3 2>/dev/null | grep -E "accuracy" | cut -c20-200 | head -4
done
