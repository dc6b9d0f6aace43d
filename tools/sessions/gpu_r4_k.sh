#!/bin/bash
# round 4, session k: HIP-graph replay of the whole forward at batch 32 (launch gaps), MLP split test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4k
O=$PWD/gpurun_out/r4k
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "encoder_split_blocks" 2>&1 | grep "rel-L1\|passed\|failed"
for g in 0 1 0 1 0 1; do ADA_GRAPH=$g python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 --warmup 5 --repeats 2 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ADA_GRAPH=$g', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', [round(v,2) for v in l['ms_per_step_repeats']], 'rel_l1', l['rel_l1'])"; done 2>&1 | tee $O/graph_b32_ab.txt
