#!/bin/bash
# round 4, session j: full GPU suite on the current tree (raw-head sub-pixel merge, new policies), configs 2 / 5, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4j
O=$PWD/gpurun_out/r4j
( time timeout 2400 python -m pytest tests -m gpu -q --durations=30 2>&1 | grep -v amdgpu | tail -n 60 ) > $O/pytest_full.txt 2>&1; tail -n 50 $O/pytest_full.txt
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu | tee $O/other_configs.txt
for sp in 0 1; do ADA_SUBPIXEL=$sp timeout 600 python tools/run_configs.py 2>&1 | grep "config 5" | sed "s/^/ADA_SUBPIXEL=$sp /"; done | tee $O/config5_subpixel_ab.txt
python bench.py > $O/bench_default_flags.json 2> $O/bench.err; tail -c 600 $O/bench_default_flags.json
