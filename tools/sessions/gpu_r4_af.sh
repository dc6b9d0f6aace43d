#!/bin/bash
# round 4, session af: heavy-tailed weights under the ssi head on ViT-L -- parity against the number of leading encoder blocks in split precision
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4af
O=$PWD/gpurun_out/r4af
KS=8,12,16,24 HEADS=auto timeout 900 python tools/enc_split_sweep.py vitl_ssi_518_heavy 2>/dev/null | tee $O/ssi_heavy_enc_split_sweep.txt
