#!/bin/bash
# round 5, session i: ladder with the token-diversity trigger and thresholds 0.42 / 0.45: kernel + model tests, parity table, qualification tool, bench with live PMC traffic
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5i
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "depth_stats or token_diversity or ladder or golden or compose or reload or graph_replay or batch32" > gpurun_out/r5i/tests.log 2>&1
tail -5 gpurun_out/r5i/tests.log
timeout 1200 python tools/parity_table.py > gpurun_out/r5i/ladder_table.txt 2>&1; cat gpurun_out/r5i/ladder_table.txt
timeout 900 python tools/qualify_checkpoint.py --encoder vitb > gpurun_out/r5i/qualify_vitb.txt 2>&1; tail -20 gpurun_out/r5i/qualify_vitb.txt
timeout 1200 python tools/qualify_checkpoint.py --encoder vitl --sizes 518x518 > gpurun_out/r5i/qualify_vitl.txt 2>&1; tail -12 gpurun_out/r5i/qualify_vitl.txt
( time timeout 900 python bench.py > gpurun_out/r5i/bench.json 2> gpurun_out/r5i/bench.err ) 2>&1 | tail -3
python -c "import json; l=json.loads(open('gpurun_out/r5i/bench.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['ms_per_step_repeats'], l['rel_l1'], l['rel_l1_low_mean'], l['low_mean']['ms_per_step'], l['roofline']['frac'], l['roofline']['traffic'], l['roofline']['traffic_source'][:60], l.get('traffic_per_family_bytes_per_launch'))"
tail -5 gpurun_out/r5i/bench.err
