#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2o
O=gpurun_out/r2o
timeout 3000 python -m pytest tests -m gpu -x -q -s > $O/test_all.log 2>&1; echo "all gpu tests rc=$?" | tee $O/summary.txt
grep -E "passed|failed|Error" $O/test_all.log | tail -n 5
timeout 900 python tools/run_configs.py 2>&1 | grep config | tee $O/other_configs.txt
timeout 300 python tools/latency_b1.py 2>&1 | tail -n 4 | tee -a $O/other_configs.txt
