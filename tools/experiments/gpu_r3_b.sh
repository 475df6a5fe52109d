#!/bin/bash
# round 3, second GPU pass: overlap probe, pair store variant A/B, new tests (aux, nms surface, multi-gpu skip)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
python -m pytest tests/test_nms.py tests/test_multi_gpu.py tests/test_hip_parity.py -m gpu -q --tb=short -rfs -x -k "nms or multi or aux or golden or pair_kernels" > $O/pytest.log 2>&1
tail -8 $O/pytest.log
python tools/overlap_probe.py --steps 30 --rounds 2 > $O/overlap.log 2>&1
grep '^{' $O/overlap.log | cut -c1-200
tail -3 $O/overlap.log | grep -v '^{'
AB_ROUNDS=2 bash tools/gpu_ab.sh r3b_ab shasta_amd/csrc/libshasta_hip.so tools/probes/_bin/libshasta_late.so
