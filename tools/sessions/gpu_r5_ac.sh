#!/bin/bash
# round 5, session ac: the raw models with their synthetic logits left un-centred (most of the ReLU map clipped to zero) -- the metric's ill-conditioned corner
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5ac
export HSA_ENABLE_IPC_MODE_LEGACY=0
PROBE_UNCENTRED=1 timeout 1500 python tools/degenerate_inputs_unbounded.py 2>&1 | grep -v amdgpu | grep "^raw\|^#\|worst" > gpurun_out/r5ac/raw_uncentred.txt; cut -c1-200 gpurun_out/r5ac/raw_uncentred.txt | tail -12
