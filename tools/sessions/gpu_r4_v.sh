#!/bin/bash
# round 4, session v: the hand-scheduled 4-wave main loop forced on every launch it supports, end to end (energy argument: it reads half the LDS bytes per FLOP)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4v
O=$PWD/gpurun_out/r4v
for i in 1 2; do
  for v in 0 16; do
    ADA_IGEMM_VARIANT=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant=$v', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'rel_l1', l['rel_l1'])"
  done
done 2>&1 | tee $O/variant_ab.txt
