#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2g
O=gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "split" -s 2>&1 | tail -n 15 | tee $O/test_split.log
timeout 2400 python -m pytest tests/test_gpu_model.py -m gpu -q -s 2>&1 | grep -E "rel-L1|passed|failed|Error|error" | tee $O/test_model.log
