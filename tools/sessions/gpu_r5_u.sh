#!/bin/bash
# round 5, session u: held-out draws for the final precision policy (first-rung projects with fp8 terms, second rung with fp8 terms, raw models with fp8 terms):
# the whole fuzz file with a fresh seed at 3x the cases -- sigmoid heads across their output range, model sizes, raw heads, igemm / attention / head operator shapes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5u
export HSA_ENABLE_IPC_MODE_LEGACY=0
ADA_FUZZ_SCALE=3 ADA_FUZZ_SEED=11 timeout 2700 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "rel-L1|passed|failed|Error" > gpurun_out/r5u/fuzz_heldout.txt
grep "rel-L1" gpurun_out/r5u/fuzz_heldout.txt | sed 's/.*rel-L1[^=]*= *//' | sort -g | tail -5; grep -E "passed|failed" gpurun_out/r5u/fuzz_heldout.txt; grep -c "second rung" gpurun_out/r5u/fuzz_heldout.txt
