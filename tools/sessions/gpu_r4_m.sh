#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4m
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -s -k "bias_per_row_group or clstoken or class_token" 2>&1 | grep -v amdgpu | grep "rel-L1\|passed\|failed\|Error\|assert" | tee gpurun_out/r4m/clstoken.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
