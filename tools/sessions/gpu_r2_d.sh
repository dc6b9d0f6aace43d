#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2d
O=gpurun_out/r2d
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -n 15 | tee $O/test_attention.log
for v in 5 3 0 5 3; do VARIANT=$v REPS=30 timeout 120 python tools/bench_attn.py 2>&1 | tail -n 1 | tee -a $O/attn_ab.txt; done
