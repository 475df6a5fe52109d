#!/bin/bash
# round 4: pair stage at F = 320 on 32-wide fp16-piece tiles (pair_f16w.hip) - the parity tests that reach it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4pw}
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_hip_parity.py -m gpu -q -x --tb=short -k "test_forward_matches_reference_golden or test_edge_shapes or test_fused_row_embeddings or test_batched_forward or test_pair_kernels or test_fp16_pair_path or test_f32_arithmetic_option or test_non_finite or test_headline_size_batches or test_forward_can_be_captured or test_fused_from_bev" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
