#!/bin/bash
# round 5, session af: the fallback the environment switch selects -- ADA_F8_CORR=0 (three fp16 terms everywhere, round-4 first rung) through the model tests and fixtures
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5af
export HSA_ENABLE_IPC_MODE_LEGACY=0
ADA_F8_CORR=0 timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_tiling.py tests/test_gpu_metrics.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 12 > gpurun_out/r5af/model_tests_fp16_terms.txt; tail -n 8 gpurun_out/r5af/model_tests_fp16_terms.txt
