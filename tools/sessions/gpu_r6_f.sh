#!/bin/bash
# round 6, session f: held-out draws of the fuzz file (ADA_FUZZ_SCALE=4, a seed the suite never runs) under the calibrated ladder: model-level families against the oracle run
# on the box (random sizes / guide types / heads, the sigmoid heads across their output range, odd sizes), kernel-level families (GEMM shapes, attention, head operators)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r6f
export HSA_ENABLE_IPC_MODE_LEGACY=0
ADA_FUZZ_SCALE=4 ADA_FUZZ_SEED=606 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -v amdgpu | grep "rel-L1\|passed\|failed\|Error\|assert" > gpurun_out/r6f/extended_fuzz.txt
tail -3 gpurun_out/r6f/extended_fuzz.txt
grep -c "rel-L1" gpurun_out/r6f/extended_fuzz.txt
grep "rel-L1 vs oracle" gpurun_out/r6f/extended_fuzz.txt | sed 's/.*rel-L1 vs oracle = \([0-9.e+-]*\).*/\1/' | sort -g | tail -5
