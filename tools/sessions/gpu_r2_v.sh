#!/bin/bash
# final round-2 profiles (after the GELU change): kernel trace of the bench command, PMC traffic passes, MFMA-busy counters, per-shape table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2v
O=$PWD/gpurun_out/r2v
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_under_rocprof.log 2>&1
ls -R $O/trace | head -n 20
DB=$(find $O/trace -name "*_results.db" | head -n 1)
[ -n "$DB" ] && python3 $R/tools/rocprof_summary.py $DB > $O/kernel_stats.csv
head -n 12 $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 1 --warmup 1 > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 1 --warmup 1 > $O/pmc_w.log 2>&1
F=$(find $O/pmc_f -name "*counter_collection.csv" | head -n 1); W=$(find $O/pmc_w -name "*counter_collection.csv" | head -n 1)
cd $R && python3 tools/pmc_traffic.py $F $W && cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 1 --warmup 1 > $O/pmc_sq.log 2>&1
find $O/pmc_sq -name "*counter_collection.csv" | head -n 2
cd $R
timeout 900 python3 tools/bench_shapes.py --batch 32 --reps 5 > $O/shapes.txt 2>&1; tail -n 3 $O/shapes.txt
# keep the merged output small: raw counter CSVs are large
find $O -name "*.db" -size +20M -delete; du -sh $O
