#!/bin/bash
# round 5, session d: GPU suite with the in-place fill + background prefetch of fixture models (durations), bench default flags
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5d
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider ) > gpurun_out/r5d/suite.log 2>&1
tail -6 gpurun_out/r5d/suite.log
grep -E "passed|failed" gpurun_out/r5d/suite.log | tail -2
timeout 600 python bench.py > gpurun_out/r5d/bench.json 2> gpurun_out/r5d/bench.err
python -c "import json; l=json.loads(open('gpurun_out/r5d/bench.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['ms_per_step_repeats'], l['rel_l1'], l['rel_l1_low_mean'], l['low_mean']['ms_per_step'], l['roofline']['frac'])"
