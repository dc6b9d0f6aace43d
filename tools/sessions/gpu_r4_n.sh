#!/bin/bash
# round 4, session n: vendor libraries at equal epilogue on today's kernels; per-tile anatomy of qkv vs fc1 (ada_debug_set_timestamps)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4n
REPS=30 timeout 600 python tools/bench_vs_lib.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4n/vs_vendor_libs.txt
timeout 600 python tools/trace_gemm.py qkv fc1_op fc1 proj_op proj 2>&1 | grep -v amdgpu | tee gpurun_out/r4n/gemm_tile_anatomy.txt
