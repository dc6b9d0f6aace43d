#!/bin/bash
# round 4, session ae: the 'ssi' (unbounded) head on ViT-L fails a new held-out fixture (1.09e-3): parity against the number of leading encoder blocks in split precision
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ae
O=$PWD/gpurun_out/r4ae
KS=0,4,8,12,16,24 HEADS=auto timeout 900 python tools/enc_split_sweep.py vitl_ssi_518 vitl_ssi_518_w1 vitl_ssi_518_w2 vitb_ssi_518 2>/dev/null | tee $O/ssi_enc_split_sweep.txt
