#!/bin/bash
# The closing profiles of a round -- ROUND=6 gpurun -- 'ROUND=6 bash tools/sessions/closing.sh' writes gpurun_out/r${ROUND}z/, whose summaries are copied to
# profiles/r0${ROUND}_z_*: full GPU suite (durations), parity of every fixture, bench with default flags (live PMC traffic), rocprofv3 kernel trace of the bench
# command, PMC traffic passes (the committed, digest-stamped fallback), SQ counters, per-shape table, configs 2 / 5 (+ kernel trace of config 2).
# (Rounds 2-5 each had their own copy of this script plus one script per experiment session: 46 files by round 5; they are in the history up to 67edbad.)
cd "$GRAFT_REPO_ROOT" || exit 1
ROUND=${ROUND:-6}
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
export ADA_COLLECTED="round $ROUND, $(date -u +%Y-%m-%dT%H:%MZ)"
mkdir -p gpurun_out/r${ROUND}z
O=$PWD/gpurun_out/r${ROUND}z
R=$PWD
( time timeout 1800 python -m pytest tests -m gpu -q --durations=0 -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -n 700 ) > $O/gpu_suite.txt 2>&1; tail -n 6 $O/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -n 2
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "golden or batch32" -p no:cacheprovider 2>&1 | grep "rel-L1" | sed 's/^\.//' > $O/parity_vs_reference_goldens.txt; wc -l $O/parity_vs_reference_goldens.txt
( time python bench.py > $O/bench_default_flags.json 2> $O/bench_default_flags.err ) 2>&1 | tail -n 3; tail -c 400 $O/bench_default_flags.json
timeout 900 python tools/run_configs.py 2>&1 | grep -v amdgpu > $O/other_configs.txt; cat $O/other_configs.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python3 $R/bench.py --no-cpu-baseline --no-traffic --no-low-mean --fixed-ladder 0.6 --steps 5 --warmup 2 > $O/bench_under_rocprof.log 2>&1
DB=$(find $O/trace -name "*_results.db" | head -n 1)
[ -n "$DB" ] && python3 $R/tools/rocprof_summary.py $DB > $O/kernel_stats.csv
head -n 12 $O/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --no-traffic --no-low-mean --fixed-ladder 0.6 --steps 1 --warmup 1 --repeats 0 > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --no-traffic --no-low-mean --fixed-ladder 0.6 --steps 1 --warmup 1 --repeats 0 > $O/pmc_w.log 2>&1
F=$(find $O/pmc_f -name "*counter_collection.csv" | head -n 1); W=$(find $O/pmc_w -name "*counter_collection.csv" | head -n 1)
cd $R && python3 tools/pmc_traffic.py $F $W && cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -o run -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --no-traffic --no-low-mean --fixed-ladder 0.6 --steps 1 --warmup 1 --repeats 0 > $O/pmc_sq.log 2>&1
S=$(find $O/pmc_sq -name "*counter_collection.csv" | head -n 1); [ -n "$S" ] && python3 $R/tools/pmc_sq_summary.py $S > $O/pmc_sq_summary.json
head -c 800 $O/pmc_sq_summary.json
cat > /tmp/cfg2.py <<'PY'
import os, sys
R = os.environ["ADA_ROOT"]
for p in (R, os.path.join(R, "amodal-depth-anything_amd")):
    sys.path.insert(0, p)
import torch
from src.models import get_model
from src.util.synth_weights import centred_final_bias, fill_state_dict_, make_inputs
m = get_model("AmodalDAv2", guide_type="mask+observation", loss_stategy="entire_target_object", encoder="vitb", pretrained=False).eval()
sd = {k: v.clone() for k, v in m.state_dict().items()}
fill_state_dict_(sd, 0)
cb = centred_final_bias("vitb", R)
sd[cb[0]] = torch.full_like(sd[cb[0]], cb[1])
m.load_state_dict(sd)
m = m.cuda()
x, _, mask, obs = make_inputs(8, 518, 518, 0, device="cuda")
with torch.no_grad():
    for _ in range(12):
        m(x, guide_mask=mask, observation=obs)
torch.cuda.synchronize()
PY
ADA_ROOT=$R ADA_LADDER_CALIBRATE=0 ADA_LADDER_R=0.6 rocprofv3 --kernel-trace --stats -d $O/trace_cfg2 -o run -- python3 /tmp/cfg2.py > $O/cfg2_under_rocprof.log 2>&1
DB2=$(find $O/trace_cfg2 -name "*_results.db" | head -n 1)
[ -n "$DB2" ] && python3 $R/tools/rocprof_summary.py $DB2 > $O/config2_kernel_stats.csv
head -n 8 $O/config2_kernel_stats.csv
cd $R
ENCODER=vitb B=8 timeout 600 python tools/config_shapes.py 2>&1 | grep -v amdgpu > $O/config2_shapes.txt; head -n 4 $O/config2_shapes.txt
timeout 900 python3 tools/bench_shapes.py --batch 32 --reps 5 2>&1 | grep -v amdgpu > $O/shapes.txt; head -n 6 $O/shapes.txt
find $O -name "*.db" -size +20M -delete; find $O -name "*counter_collection.csv" -size +20M -delete; du -sh $O
