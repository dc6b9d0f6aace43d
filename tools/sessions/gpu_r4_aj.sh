#!/bin/bash
# round 4, session aj: the four 1x1 projects with weight-only split precision ("projw": 2x their MACs) against the full split ("proj": 3x) and none
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4aj
O=$PWD/gpurun_out/r4aj
export SUBSETS="out1,out2,out3;out1,out2,out3,projw;out1,out2,out3,proj"
timeout 1200 python tools/head_split_sweep.py vitb_714x1022_heavy vitl_714x1022_heavy vitl_518_heavy vitl_518 vitb_518_heavy vitb_518_b8_struct vitl_518_struct vitl_518_b8 2>/dev/null | tee $O/projw.txt
for i in 1 2; do for hs in "out1,out2,out3" "out1,out2,out3,projw"; do
  ADA_HEAD_SPLIT=$hs python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head=$hs', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'rel_l1', l['rel_l1'])"
done; done | tee $O/projw_bench.txt
