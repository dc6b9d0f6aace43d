#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2i
O=gpurun_out/r2i
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -s -k "fold or split" 2>&1 | tail -n 25 | tee $O/test_fold.log
timeout 2400 python -m pytest tests/test_gpu_model.py -m gpu -q -s 2>&1 | grep -E "rel-L1|passed|failed|Error|error|grey" | tee $O/test_model.log
timeout 900 python bench.py --no-cpu-baseline 2>&1 | tail -n 1 | tee $O/bench_fold.json
