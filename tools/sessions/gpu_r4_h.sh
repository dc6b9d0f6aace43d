#!/bin/bash
# round 4, session h: output_conv1 commute with fp32 / operand-typed tap maps vs the old path: speed and parity (heavy / struct / batch-8 fixtures)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4h
O=$PWD/gpurun_out/r4h
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tapsum" 2>&1 | grep -v amdgpu | tail -n 3
for c in 0 16 32; do
  echo "== ADA_OC1_COMMUTE=$c"
  ADA_OC1_COMMUTE=$c timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and (vitb or vitl) and not bench" 2>&1 | grep "rel-L1\|passed\|failed"
done 2>&1 | tee $O/oc1_commute_parity.txt
for c in 0 16 32 0 16 32; do ADA_OC1_COMMUTE=$c python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OC1_COMMUTE=$c', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', l['ms_per_step_repeats'], 'rel_l1', l['rel_l1'], 'igemm frac', round(l['roofline']['frac'],4))"; done 2>&1 | tee $O/oc1_commute_ab.txt
