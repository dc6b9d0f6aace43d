#!/bin/bash
# round 5, session l: ViT-L second rung without the two finest ResidualConvUnit levels: ladder tests, the ViT-L low-mean / constant fixtures, bench (low_mean cost)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5l
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -s -p no:cacheprovider -k "ladder or vitl_518_m10 or vitl_518_struct_m20 or vitl_518_zeros or bench_vitl_b32_low or vitb_518_m20" 2>&1 | grep -E "rel-L1|passed|failed|r = " > gpurun_out/r5l/tests.log
cat gpurun_out/r5l/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-traffic --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['rel_l1'], l['rel_l1_low_mean'], l['low_mean'])"
