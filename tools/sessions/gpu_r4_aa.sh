#!/bin/bash
# round 4, session aa: staggered start of the first round of tiles (do de-synchronised epilogues relieve the store path?)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4aa
O=$PWD/gpurun_out/r4aa
for st in 0 1 2 4 8; do
  ADA_IGEMM_STAGGER=$st timeout 600 python tools/bench_shapes.py --reps 5 2>/dev/null | grep '"igemm"' | grep '"M": 43840' | sed "s/^/stagger=$st /"
done | tee $O/stagger_shapes.txt
for st in 0 2 4 0 2 4; do
  ADA_IGEMM_STAGGER=$st python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger=$st', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms')"
done | tee $O/stagger_ab.txt
