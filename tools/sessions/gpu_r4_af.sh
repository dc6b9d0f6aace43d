#!/bin/bash
# round 4, session af: heavy-tailed weights under the unbounded heads -- parity against the number of leading encoder blocks in split precision
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4af
O=$PWD/gpurun_out/r4af
KS=0,2,4,6,8,12 HEADS=auto timeout 1200 python tools/enc_split_sweep.py raw_vitb_518_heavy raw_vitb_518_heavy_w1 raw_vitb_518 raw_vitl_518_heavy raw_vitl_518_heavy_w1 raw_vitl_518 2>/dev/null | tee $O/raw_heavy_enc_split_sweep.txt
