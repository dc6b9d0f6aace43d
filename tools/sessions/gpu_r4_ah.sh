#!/bin/bash
# round 4, session ah: every raw / ssi fixture and the whole-model fuzz with the generalised unbounded-head policy (first 4 / 8 encoder blocks in split precision)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ah
O=$PWD/gpurun_out/r4ah
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "test_hip_forward_matches_reference_golden and (raw or ssi)" 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' | tee $O/unbounded_fixtures.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_model.py -q -m gpu -x -k "not test_hip_forward_matches_reference_golden" 2>&1 | tail -3 | tee -a $O/unbounded_fixtures.txt
