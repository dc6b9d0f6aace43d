#!/bin/bash
# round 4, session e: raw ViT-G -- split precision in the first K encoder blocks x head policy, all raw ViT-G fixtures, config 5 timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4e
O=$PWD/gpurun_out/r4e
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "weight_only or attention" 2>&1 | grep -v amdgpu | tail -n 5
timeout 2400 python tools/enc_split_sweep.py --time 2>&1 | grep -v amdgpu | tee $O/enc_split_sweep.txt
