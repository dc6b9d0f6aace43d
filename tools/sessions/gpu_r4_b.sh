#!/bin/bash
# round 4, session b: sub-pixel conv + LN second output kernel tests, full parity table (all fixtures, no -x), A/B bench ADA_SUBPIXEL=0/1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4b
O=$PWD/gpurun_out/r4b
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "subpixel or second_output or layernorm or conv3x3 or conv_transpose" 2>&1 | grep -v amdgpu | tail -n 30 > $O/pytest_kernels.txt; tail -n 12 $O/pytest_kernels.txt
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32 or folding or module_by_module" 2>&1 | grep -v amdgpu > $O/pytest_model.txt; grep "rel-L1\|passed\|failed\|Error" $O/pytest_model.txt
for sp in 0 1 0 1; do ADA_SUBPIXEL=$sp python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SUBPIXEL=$sp', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', l['ms_per_step_repeats'], 'rel_l1', l['rel_l1'], 'igemm frac', round(l['roofline']['frac'],4))"; done 2>&1 | tee $O/subpixel_ab.txt
