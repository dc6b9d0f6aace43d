#!/bin/bash
# round 5, session a: GPU suite with per-test durations (new off-centre fixtures + model cache), ladder table, bench default flags
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5a
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q --durations=60 -p no:cacheprovider ) > gpurun_out/r5a/suite.log 2>&1
tail -5 gpurun_out/r5a/suite.log
timeout 900 python tools/parity_table.py > gpurun_out/r5a/ladder_table.txt 2>&1
cat gpurun_out/r5a/ladder_table.txt | tail -40
timeout 600 python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
tail -c 3000 gpurun_out/r5a/bench.json
