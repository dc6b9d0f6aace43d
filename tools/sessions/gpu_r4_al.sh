#!/bin/bash
# round 4, session al: do the composed-weight launches (sub-pixel merges, commuted output_conv1) carry the sigmoid stress fixtures' error?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4al
O=$PWD/gpurun_out/r4al
for env in "" "ADA_SUBPIXEL=0" "ADA_OC1_COMMUTE=0" "ADA_SUBPIXEL=0 ADA_OC1_COMMUTE=0"; do
  env $env timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "714x1022_heavy or vitb_518_heavy or vitb_518_b8_struct" 2>&1 | grep "rel-L1" | sed "s/^\.//;s/^/[$env] /"
done | tee $O/composed_weights.txt
