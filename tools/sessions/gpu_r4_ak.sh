#!/bin/bash
# round 4, session ak: what the margin knobs of the sigmoid ViT-L model cost at the benchmark (bs=32, 518^2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ak
O=$PWD/gpurun_out/r4ak
run() { python bench.py --no-cpu-baseline --steps 10 --warmup 3 --repeats 0 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'rel_l1', '%.3e' % l['rel_l1'])"; }
( run "default (out1,out2,out3)"
  ADA_HEAD_SPLIT=out1,out2,out3,projw run "out1,out2,out3,projw"
  ADA_HEAD_SPLIT=out1,out2,out3,projw,oc2 run "out1,out2,out3,projw,oc2"
  ADA_HEAD_SPLIT=split run "head split"
  ADA_HEAD_SPLIT=split ADA_ENC_SPLIT=8 run "head split + first 8 encoder blocks"
  run "default (out1,out2,out3)" ) | tee $O/knob_costs.txt
SHIFTS=-3,0 timeout 600 python - <<'PY' 2>/dev/null | tee -a $O/knob_costs.txt
import os, sys
sys.argv = ["operating_point.py", "vitl_518"]
sys.path.insert(0, "tools")
import operating_point as op
op.POLICIES = [("auto", "auto"), ("out1,out2,out3,projw,oc2", "auto"), ("split", "auto")]
op.main()
PY
