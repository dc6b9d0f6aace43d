#!/bin/bash
# round 5, session x: ViT-S (whole head in split precision, no ladder) at the bottom of the sigmoid's range
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5x
export HSA_ENABLE_IPC_MODE_LEGACY=0
PROBE_VITS=1 timeout 900 python tools/third_rung_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r5x/vits_probe.txt; cat gpurun_out/r5x/vits_probe.txt
