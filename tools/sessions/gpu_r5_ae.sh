#!/bin/bash
# round 5, session ae: third set of held-out draws (ADA_FUZZ_SCALE=3 ADA_FUZZ_SEED=31, the whole fuzz file) on the final policy
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5ae
export HSA_ENABLE_IPC_MODE_LEGACY=0
ADA_FUZZ_SCALE=3 ADA_FUZZ_SEED=31 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "rel-L1|passed|failed|Error" > gpurun_out/r5ae/fuzz_heldout3.txt
grep "rel-L1" gpurun_out/r5ae/fuzz_heldout3.txt | sed 's/.*rel-L1[^=]*= *//' | sort -g | tail -5; grep -E "passed|failed" gpurun_out/r5ae/fuzz_heldout3.txt; grep -c "second rung" gpurun_out/r5ae/fuzz_heldout3.txt; grep -c "third rung" gpurun_out/r5ae/fuzz_heldout3.txt
