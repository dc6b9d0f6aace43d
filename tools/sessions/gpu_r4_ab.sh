#!/bin/bash
# round 4, session ab: staggered start, end to end, five alternating rounds of 40 steps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ab
O=$PWD/gpurun_out/r4ab
for i in 1 2 3 4 5; do
  for st in 0 1 2; do
    ADA_IGEMM_STAGGER=$st python bench.py --no-cpu-baseline --steps 40 --warmup 5 --repeats 0 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger=$st', round(l['value'],1), 'img/s', round(l['ms_per_step'],3), 'ms')"
  done
done | tee $O/stagger_ab5.txt
