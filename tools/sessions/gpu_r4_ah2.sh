#!/bin/bash
# round 4, session ah2: whole-model fuzz and the model tests with the generalised unbounded-head policy
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ah
O=$PWD/gpurun_out/r4ah
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_model.py tests/test_gpu_tiling.py tests/test_gpu_metrics.py -q -m gpu -k "not test_hip_forward_matches_reference_golden" 2>&1 | grep "passed\|failed\|Error\|assert" | head -20 | tee $O/other_model_tests.txt
