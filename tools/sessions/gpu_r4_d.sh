#!/bin/bash
# round 4, session d: attention VALU trim A/B against the round-3 kernel (same box), column-group width A/B end to end, full GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4d
O=$PWD/gpurun_out/r4d
R=$PWD
# the round-3 attention kernel, built into a second library from the same tree
mkdir -p /tmp/x/amodal-depth-anything_amd /tmp/x/tools && cp -r amodal-depth-anything_amd/csrc /tmp/x/amodal-depth-anything_amd/ && cp -r include /tmp/x/ && cp tools/isa_guard.py /tmp/x/tools/
cp tools/scratch/ada_attention_r3.hip /tmp/x/amodal-depth-anything_amd/csrc/ada_attention.hip
rm -f /tmp/x/amodal-depth-anything_amd/csrc/*.so /tmp/x/amodal-depth-anything_amd/csrc/*.stamp
python /tmp/x/amodal-depth-anything_amd/csrc/build.py > $O/build_r3attn.log 2>&1; OLD=/tmp/x/amodal-depth-anything_amd/csrc/libada_hip.so; ls -la $OLD
for i in 1 2 3; do
  ADA_HIP_LIB=$OLD REPS=50 python tools/bench_attn.py 2>/dev/null | sed 's/^/r3 kernel   /'
  REPS=50 python tools/bench_attn.py 2>/dev/null | sed 's/^/r4 kernel   /'
done | tee $O/attention_ab_isolated.txt
for i in 1 2; do
  for lib in $OLD ""; do
    ADA_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-r4}', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'attn us', round(1e3*l['roofline_attention']['avg_launch_ms'],1), 'attn frac', round(l['roofline_attention']['frac'],4), 'igemm frac', round(l['roofline']['frac'],4))"
  done
done 2>&1 | tee $O/attention_ab_end_to_end.txt
for g in 0 6 8 0 6 8; do
  ADA_IGEMM_GROUP=$g python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GROUP=$g', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'igemm frac', round(l['roofline']['frac'],4), 'rel_l1', l['rel_l1'])"
done 2>&1 | tee $O/group_ab_end_to_end.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 2>&1 | grep -v amdgpu | tail -n 45 ) > $O/pytest_full.txt 2>&1; tail -n 40 $O/pytest_full.txt
