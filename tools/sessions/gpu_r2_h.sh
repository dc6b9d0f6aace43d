#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2h
O=gpurun_out/r2h
timeout 2700 python -m pytest tests -m gpu -x -q -s > $O/test_all.log 2>&1; echo "all gpu tests rc=$?" | tee $O/summary.txt
grep -E "rel-L1|passed|failed|Error|within 1 grey|split GEMM" $O/test_all.log | tail -n 45
timeout 900 python bench.py 2>&1 | tail -n 1 | tee $O/bench_default.json
ADA_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 900 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -n 1 | tee $O/bench_forced_dist.json
