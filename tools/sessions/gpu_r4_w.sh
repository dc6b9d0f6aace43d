#!/bin/bash
# round 4, session w: one-off extended fuzz sweep on the final kernels, fresh seeds, 4x the cases of every family
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4w
O=$PWD/gpurun_out/r4w
ADA_FUZZ_SCALE=4 ADA_FUZZ_SEED=4004 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -s -x 2>&1 | grep -v "^$" > $O/fuzz_full.txt
tail -3 $O/fuzz_full.txt
grep "rel-L1" $O/fuzz_full.txt | sed 's/.*rel-L1 vs oracle = //' | sort -g | tail -3
