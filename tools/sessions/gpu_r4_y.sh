#!/bin/bash
# round 4, session y: register-direct epilogue of the 256x256 tile (operands swapped, no LDS transpose) -- kernel tests, fuzz, per-shape and end-to-end A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4y
O=$PWD/gpurun_out/r4y
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -8 | tee $O/pytest_direct.txt
for v in 0 1; do
  ADA_IGEMM_DIRECT=$v timeout 600 python tools/bench_shapes.py --reps 5 > $O/shapes_direct$v.txt 2>&1
done
for i in 1 2; do
  for v in 0 1; do
    ADA_IGEMM_DIRECT=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('direct=$v', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'rel_l1', l['rel_l1'])"
  done
done 2>&1 | tee $O/direct_ab.txt
