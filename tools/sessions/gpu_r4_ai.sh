#!/bin/bash
# round 4, session ai: which head groups would buy the sigmoid ViT-B / ViT-L models margin on the heavy-tailed large-resolution fixtures (8.9e-4 / 9.0e-4)?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4ai
O=$PWD/gpurun_out/r4ai
export SUBSETS="out1,out2,out3;out1,out2,out3,proj;out1,out2,out3,tok;out1,out2,out3,ip;out1,out2,out3,rn;out1,out2,out3,oc2;out1,out2,out3,oc1;out1,out2,out3,rcu;out1,out2,out3,tok,ip,rn;tok,ip,rn,rcu,out,oc1,oc2"
timeout 1200 python tools/head_split_sweep.py vitb_714x1022_heavy vitl_714x1022_heavy vitl_518_heavy vitl_518 2>/dev/null | tee $O/sigmoid_head_groups.txt
