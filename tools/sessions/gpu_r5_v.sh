#!/bin/bash
# round 5, session v: the bottom of the sigmoid heads' output range under wider policies (tools/third_rung_probe.py) -- what a third rung would have to be
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5v
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python tools/third_rung_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r5v/third_rung_probe.txt; cat gpurun_out/r5v/third_rung_probe.txt
