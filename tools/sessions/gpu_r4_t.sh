#!/bin/bash
# round 4, session t: row peeling of a nearly empty last round (qkv of ViT-L bs=32: 2064 tiles = 8 rounds + 16) -- kernel tests, then end to end A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4t
O=$PWD/gpurun_out/r4t
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "peels or forced_tile or swiglu" 2>&1 | tail -5 | tee $O/pytest_peel.txt
for i in 1 2 3; do
  for v in 0 1; do
    ADA_IGEMM_PEEL=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('peel=$v', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'rel_l1', l['rel_l1'])"
  done
done 2>&1 | tee $O/peel_ab.txt
