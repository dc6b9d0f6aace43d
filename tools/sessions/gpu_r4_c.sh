#!/bin/bash
# round 4, session c: where is the raw ViT-G error -- head split subsets on the new robustness fixtures
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4c
O=$PWD/gpurun_out/r4c
SUBSETS="oc2,out,rn1,rn2,rn3,proj,rs1,rs3;oc1,oc2,out,rn,proj,rs0,rs1,rs3;tok,ip,rn,out,oc1,oc2;tok,ip,rn,rcu,out,oc1,oc2" timeout 1200 python tools/head_split_sweep.py raw_vitg_224 raw_vitg_224_w1 raw_vitg_224_w2 raw_vitg_224_heavy raw_vitg_224_struct 2>&1 | grep -v amdgpu | tee $O/head_split_new_fixtures.txt
