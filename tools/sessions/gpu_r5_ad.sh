#!/bin/bash
# round 5, session ad: input-side trigger only for the heads without a sigmoid -- the rung tests, the unbounded fixtures, the probe, configs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5ad
O=gpurun_out/r5ad
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "ladder or flat_input or (golden and (raw or ssi))" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed\|Error" | sed 's/^\.//' > $O/tests.txt; sort -t= -k2 -g $O/tests.txt | tail -n 4; tail -n 1 $O/tests.txt
timeout 1500 python tools/degenerate_inputs_unbounded.py 2>&1 | grep -v amdgpu > $O/degenerate_unbounded.txt; grep -c "re-run" $O/degenerate_unbounded.txt; grep "noise" $O/degenerate_unbounded.txt | grep -c "re-run"; tail -n 1 $O/degenerate_unbounded.txt
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu
