#!/bin/bash
# round 4, session i: sigmoid ViT-B / ViT-L heads with the 1x1 out_convs of levels 1-3 in split precision: parity and cost
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4i
O=$PWD/gpurun_out/r4i
for hs in "" "out1,out2,out3" "out1,out2,out3,proj"; do
  echo "== ADA_HEAD_SPLIT=[$hs]"
  if [ -z "$hs" ]; then unset ADA_HEAD_SPLIT; else export ADA_HEAD_SPLIT=$hs; fi
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and (vitb or vitl) and not bench" 2>&1 | grep "rel-L1\|passed\|failed"
  for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'rel_l1', l['rel_l1'])"; done
done 2>&1 | tee $O/head_out_split.txt
