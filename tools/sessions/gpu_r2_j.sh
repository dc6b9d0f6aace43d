#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2j
O=gpurun_out/r2j
for i in 1 2; do for f in 1 0; do
  echo "ADA_FOLD_LN=$f" | tee -a $O/fold_ab.txt
  ADA_FOLD_LN=$f timeout 600 python bench.py --no-cpu-baseline --steps 15 --warmup 4 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), 'igemm ms/step', round(d['roofline']['avg_launch_ms']*d['roofline']['launches']/d['steps'],2), 'attn', round(d['roofline_attention']['avg_launch_ms'],4), 'rel_l1', d['rel_l1'])" | tee -a $O/fold_ab.txt
done; done
