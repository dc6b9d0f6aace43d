#!/bin/bash
# round 4, session o: LayerNorm tail (last-arriving tile normalises its row panel): kernel tests, model parity, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4o
O=$PWD/gpurun_out/r4o
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "layernorm_tail or forced_tile_epilogues or conv3x3 or subpixel" 2>&1 | grep -v amdgpu | tail -n 8
for t in 0 1; do
  echo "== ADA_LN_TAIL=$t"
  ADA_LN_TAIL=$t timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and (vitb_518 or vitl_518 or vits_518 or raw_vitg_224_w1) and not b8" 2>&1 | grep "rel-L1\|passed\|failed"
done 2>&1 | tee $O/ln_tail_parity.txt
for t in 0 1 0 1 0 1; do ADA_LN_TAIL=$t python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LN_TAIL=$t', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', l['ms_per_step_repeats'], 'rel_l1', l['rel_l1'], 'igemm frac', round(l['roofline']['frac'],4), 'exec', round(l['roofline']['frac_executed'],4))"; done 2>&1 | tee $O/ln_tail_ab.txt
