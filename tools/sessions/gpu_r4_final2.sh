#!/bin/bash
# round-4 closing session on the final tree (csrc unchanged since gpu_r4_final.sh; tests, fixtures and the precision policy of the unbounded heads changed):
# full GPU suite, parity table over all reference fixtures, bench with default flags
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4z2
O=$PWD/gpurun_out/r4z2
( time timeout 1800 python -m pytest tests -m gpu -q --durations=14 2>&1 | grep -v amdgpu | tail -n 27 ) > $O/gpu_suite.txt 2>&1; tail -n 6 $O/gpu_suite.txt
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32 or config5" 2>&1 | grep "rel-L1" | sed 's/^\.//;s/^s//' > $O/parity_vs_reference_goldens.txt; wc -l $O/parity_vs_reference_goldens.txt
python bench.py > $O/bench_default_flags.json 2> $O/bench_default_flags.err; tail -c 400 $O/bench_default_flags.json
