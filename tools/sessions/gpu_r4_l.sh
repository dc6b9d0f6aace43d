#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4l
timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -k "odd_sizes" 2>&1 | grep -v amdgpu | grep "rel-L1\|passed\|failed\|Error\|assert" | tee gpurun_out/r4l/odd_sizes.txt
