#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c
for v in 0 11 12 13 14 3; do VARIANT=$v REPS=30 timeout 120 python tools/bench_attn.py 2>&1 | tail -n 1 | tee -a $O/attn_ablation.txt; done
