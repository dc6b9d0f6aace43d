#!/bin/bash
# round 5, session q: where the fp8 correction terms cost parity margin -- encoder-only / head-only A/B on the 'ssi', raw and heavy-tailed fixtures; config 5 launch by launch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5q
O=gpurun_out/r5q
export HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in 0 enc head 1; do
  ADA_F8_CORR=$mode timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden and (ssi or raw or heavy)" -p no:cacheprovider 2>&1 | grep "rel-L1\|passed\|failed" | sed 's/^\.//' > $O/parity_mode_$mode.txt; tail -n 1 $O/parity_mode_$mode.txt
done
RAW=1 ENCODER=vitg B=8 SIZE=1022 REPS=3 timeout 1200 python tools/config_shapes.py > $O/config5_shapes.txt 2>&1; head -n 8 $O/config5_shapes.txt
