#!/bin/bash
# round 4, session am: what wider head policies would cost the sigmoid ViT-B model at config 2 (bs=8, 518^2), and what they give on its fixtures
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4am
O=$PWD/gpurun_out/r4am
for hs in "out1,out2,out3" "out1,out2,out3,projw" "out1,out2,out3,tok" "out1,out2,out3,rcu" "out1,out2,out3,tok,ip,rn" "out1,out2,out3,tok,rcu"; do
  for i in 1 2; do ADA_HEAD_SPLIT=$hs timeout 300 python tools/run_configs.py 2>/dev/null | grep "config 2" | sed "s/^/[$hs] /"; done
done 2>&1 | grep -v "^$" | tee $O/vitb_policy_cost.txt
export SUBSETS="out1,out2,out3;out1,out2,out3,tok;out1,out2,out3,rcu;out1,out2,out3,tok,rcu;out1,out2,out3,tok,ip,rn"
timeout 900 python tools/head_split_sweep.py vitb_714x1022_heavy vitb_518_heavy vitb_518_b8_struct vitb_518_b8_w1 vitb_1022 2>/dev/null | tee $O/vitb_policy_parity.txt
