#!/bin/bash
# round 5, session aa: the models without a sigmoid on constant / checkerboard inputs (tools/degenerate_inputs_unbounded.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5aa
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python tools/degenerate_inputs_unbounded.py 2>&1 | grep -v amdgpu > gpurun_out/r5aa/degenerate_unbounded.txt; cat gpurun_out/r5aa/degenerate_unbounded.txt
