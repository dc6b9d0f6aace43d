#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "dpt_tail" 2>&1 | tail -n 15 | tee $O/test_tail.log
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or variants" 2>&1 | grep -E "rel-L1|passed|failed|Error|error" | tee $O/test_model.log
for f in 1 0 1 0; do
  echo "ADA_FUSED_TAIL=$f" | tee -a $O/tail_ab.txt
  ADA_FUSED_TAIL=$f timeout 600 python bench.py --no-cpu-baseline --steps 15 --warmup 4 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), 'rel_l1', d['rel_l1'])" | tee -a $O/tail_ab.txt
done
