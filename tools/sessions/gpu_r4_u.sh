#!/bin/bash
# round 4, session u: the qkv launch alone with and without row peeling
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4u
O=$PWD/gpurun_out/r4u
for v in 0 1 0 1; do
  ADA_IGEMM_PEEL=$v timeout 600 python tools/bench_shapes.py --reps 10 2>/dev/null | grep '"N": 3072' | sed "s/^/peel=$v /"
done | tee $O/qkv_peel.txt
