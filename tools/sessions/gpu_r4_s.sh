#!/bin/bash
# round 4, session s: tile 5 (256x128x32, 4 waves, two workgroups per CU) -- parity of every forced-tile test, then per-shape times against the default tiles
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4s
O=$PWD/gpurun_out/r4s
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "forced_tile" 2>&1 | tail -5 | tee $O/pytest_tile5.txt
timeout 600 python tools/bench_shapes.py --reps 5 > $O/shapes_default.txt 2>&1
ADA_IGEMM_TILE=5 timeout 600 python tools/bench_shapes.py --reps 5 > $O/shapes_tile5.txt 2>&1
tail -3 $O/shapes_tile5.txt
