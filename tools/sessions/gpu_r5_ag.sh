#!/bin/bash
# round 5, session ag: more weight draws of the raw / 'ssi' models on plain inputs (PROBE_DRAWS=1 tools/degenerate_inputs_unbounded.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5ag
export HSA_ENABLE_IPC_MODE_LEGACY=0
PROBE_DRAWS=1 timeout 2400 python tools/degenerate_inputs_unbounded.py 2>&1 | grep -v amdgpu > gpurun_out/r5ag/unbounded_draws.txt; grep "rel-L1" gpurun_out/r5ag/unbounded_draws.txt | sort -t= -k2 -g | tail -8 | cut -c1-170; tail -n 1 gpurun_out/r5ag/unbounded_draws.txt
