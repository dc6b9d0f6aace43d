#!/bin/bash
# round 5, session h: what the ladder's second rung needs (head group subsets on the low-mean / constant fixtures), token diversity of constant inputs, parity table with the new fixtures
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/r5h
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python tools/ladder_subsets.py vitl_518_m10 bench_vitl_b32_low vitl_518_struct_m20 vitl_518_zeros_c vitl_518_zeros > gpurun_out/r5h/subsets_vitl.txt 2>&1; cat gpurun_out/r5h/subsets_vitl.txt
timeout 1500 python tools/ladder_subsets.py vitb_518_zeros vitb_126x154_zeros vitb_518_struct_m10 vitb_518_m20 vitb_518_zeros_c vitb_266x322_struct_m30 vitb_518_checker > gpurun_out/r5h/subsets_vitb.txt 2>&1; cat gpurun_out/r5h/subsets_vitb.txt
timeout 1200 python tools/parity_table.py > gpurun_out/r5h/ladder_table.txt 2>&1; cat gpurun_out/r5h/ladder_table.txt
