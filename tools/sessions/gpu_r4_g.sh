#!/bin/bash
# round 4, session g: tap-sum resize kernel test, goldens through the commuted output_conv1, A/B ADA_OC1_COMMUTE, raw ViT-G under the new policy (incl. held-out w3 / w4), config 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4g
O=$PWD/gpurun_out/r4g
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -s -k "tapsum" 2>&1 | grep -v amdgpu | tail -n 15
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32 or config5" 2>&1 | grep -v amdgpu > $O/pytest_model.txt; grep "rel-L1\|passed\|failed\|Error" $O/pytest_model.txt
for c in 0 1 0 1; do ADA_OC1_COMMUTE=$c python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OC1_COMMUTE=$c', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', l['ms_per_step_repeats'], 'rel_l1', l['rel_l1'], 'igemm frac', round(l['roofline']['frac'],4))"; done 2>&1 | tee $O/oc1_commute_ab.txt
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu | tee $O/other_configs.txt
timeout 600 python tools/bench_shapes.py --batch 32 --reps 5 2>&1 | grep -v amdgpu > $O/shapes.txt; head -n 30 $O/shapes.txt
