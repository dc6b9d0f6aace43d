#!/bin/bash
# round 5, session g: revised tile estimate + compose_f32 + quarter-wave LN: whole GPU suite, config 2 / 5 numbers, bench default flags, checkpoint qualification tool
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5g
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider ) > gpurun_out/r5g/suite.log 2>&1
grep -E "passed|failed|^FAILED|^real" gpurun_out/r5g/suite.log | tail -8
ENCODER=vitb B=8 timeout 600 python tools/config_shapes.py > gpurun_out/r5g/config2_shapes.txt 2>&1
head -3 gpurun_out/r5g/config2_shapes.txt
timeout 600 python tools/latency_b1.py vitb > gpurun_out/r5g/latency_b1.txt 2>&1; cat gpurun_out/r5g/latency_b1.txt
timeout 900 python tools/run_configs.py > gpurun_out/r5g/other_configs.txt 2>&1; cat gpurun_out/r5g/other_configs.txt
timeout 600 python bench.py > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.err
python -c "import json; l=json.loads(open('gpurun_out/r5g/bench.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['ms_per_step_repeats'], l['rel_l1'], l['rel_l1_low_mean'], l['low_mean']['ms_per_step'], l['roofline']['frac'])"
timeout 900 python tools/qualify_checkpoint.py --encoder vitb > gpurun_out/r5g/qualify_vitb.txt 2>&1; tail -30 gpurun_out/r5g/qualify_vitb.txt
