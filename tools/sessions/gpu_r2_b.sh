#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
timeout 300 python tools/prof_attn.py 2>&1 | grep -v amdgpu.ids | tee $O/prof_attn.txt
timeout 300 python tools/prof_gemm.py 2>&1 | grep -v amdgpu.ids | tee $O/prof_gemm.txt
