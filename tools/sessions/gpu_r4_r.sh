#!/bin/bash
# round 4, session r: write-through (sc1) stores of the fp32 residual stream in the proj / fc2 epilogues, end to end; LayerNorm tail (optimised) end to end
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4r
O=$PWD/gpurun_out/r4r
WT=$PWD/amodal-depth-anything_amd/csrc/libada_hip_epiwt.so
for i in 1 2 3; do
  for v in "plain" "wt" "lntail"; do
    if [ $v = wt ]; then export ADA_HIP_LIB=$WT; else unset ADA_HIP_LIB; fi
    if [ $v = lntail ]; then export ADA_LN_TAIL=1; else unset ADA_LN_TAIL; fi
    python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'rel_l1', l['rel_l1'], 'igemm frac', round(l['roofline']['frac'],4), 'exec', round(l['roofline']['frac_executed'],4))"
  done
done 2>&1 | tee $O/residual_store_policy_ab.txt
