#!/bin/bash
# round 4, session ag: the same sweep on ViT-S (unbounded heads, heavy-tailed and normal weights)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ag
O=$PWD/gpurun_out/r4ag
KS=0,2,4,12 HEADS=auto timeout 1200 python tools/enc_split_sweep.py raw_vits_518_heavy vits_ssi_518_heavy vits_ssi_image_mask vits_ssi_518 raw_vits_518 vitb_ssi_518_heavy vitb_ssi_518 2>/dev/null | tee $O/vits_enc_split_sweep.txt
