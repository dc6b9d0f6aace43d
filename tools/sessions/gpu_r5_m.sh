#!/bin/bash
# round 5, session m: the sigmoid heads across their output range against the oracle on the box (random operating points / styles / sizes), x4 cases
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5m
export HSA_ENABLE_IPC_MODE_LEGACY=0
ADA_FUZZ_SCALE=4 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -p no:cacheprovider -k "across_the_output_range" 2>&1 | grep -E "rel-L1|passed|failed|Error" > gpurun_out/r5m/range_fuzz.txt
sort -t= -k2 -g gpurun_out/r5m/range_fuzz.txt | tail -15; grep -E "passed|failed" gpurun_out/r5m/range_fuzz.txt
