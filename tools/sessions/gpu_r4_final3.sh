#!/bin/bash
# round-4 closing session, part 2 (final kernels, final precision policy): rocprofv3 kernel trace of the bench command and the per-shape table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4z3
O=$PWD/gpurun_out/r4z3
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/trace -name "*_results.db" | head -n 1)
[ -n "$DB" ] && python3 $R/tools/rocprof_summary.py $DB > $O/kernel_stats.csv
head -n 8 $O/kernel_stats.csv
cd $R
timeout 900 python3 tools/bench_shapes.py --batch 32 --reps 5 2>&1 | grep -v amdgpu > $O/shapes.txt; head -n 4 $O/shapes.txt
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/other_configs.txt; cat $O/other_configs.txt
find $O -name "*.db" -size +20M -delete; du -sh $O
