#!/bin/bash
# round-4 profiles: full GPU suite, parity table, bench with default flags, rocprofv3 kernel trace of the bench command, PMC traffic passes (separate,
# stamped with the csrc digest: bench.py refuses to quote traffic collected on other kernels), SQ counters, per-shape table, configs 2 and 5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export ADA_COLLECTED="round 4, $(date -u +%Y-%m-%dT%H:%MZ)"
mkdir -p gpurun_out/r4z
O=$PWD/gpurun_out/r4z
R=$PWD
( time timeout 1800 python -m pytest tests -m gpu -q --durations=12 2>&1 | grep -v amdgpu | tail -n 25 ) > $O/gpu_suite.txt 2>&1; tail -n 6 $O/gpu_suite.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32" 2>&1 | grep "rel-L1" | sed 's/^\.//' > $O/parity_vs_reference_goldens.txt; wc -l $O/parity_vs_reference_goldens.txt
python bench.py > $O/bench_default_flags.json 2> $O/bench_default_flags.err; tail -c 300 $O/bench_default_flags.json
timeout 600 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/other_configs.txt; cat $O/other_configs.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/trace -name "*_results.db" | head -n 1)
[ -n "$DB" ] && python3 $R/tools/rocprof_summary.py $DB > $O/kernel_stats.csv
head -n 12 $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 1 --warmup 1 --repeats 0 > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 1 --warmup 1 --repeats 0 > $O/pmc_w.log 2>&1
F=$(find $O/pmc_f -name "*counter_collection.csv" | head -n 1); W=$(find $O/pmc_w -name "*counter_collection.csv" | head -n 1)
cd $R && python3 tools/pmc_traffic.py $F $W && cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 1 --warmup 1 --repeats 0 > $O/pmc_sq.log 2>&1
S=$(find $O/pmc_sq -name "*counter_collection.csv" | head -n 1); [ -n "$S" ] && python3 $R/tools/pmc_sq_summary.py $S > $O/pmc_sq_summary.json
head -c 1200 $O/pmc_sq_summary.json
cd $R
python bench.py --no-cpu-baseline --repeats 0 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('traffic after refresh:', l['roofline']['traffic'], l['roofline']['traffic_source'])"
timeout 900 python3 tools/bench_shapes.py --batch 32 --reps 5 2>&1 | grep -v amdgpu > $O/shapes.txt; head -n 6 $O/shapes.txt
find $O -name "*.db" -size +20M -delete; find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O
