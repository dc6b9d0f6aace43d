#!/bin/bash
# round 4, session p: anatomy of the LayerNorm tail; A/B of the k-walk scratch fix against HEAD's igemm on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4p
O=$PWD/gpurun_out/r4p
timeout 300 python tools/bench_ln_tail.py 2>&1 | grep -v amdgpu | tee $O/ln_tail_anatomy.txt
mkdir -p /tmp/x/amodal-depth-anything_amd /tmp/x/tools && cp -r amodal-depth-anything_amd/csrc /tmp/x/amodal-depth-anything_amd/ && cp -r include /tmp/x/ && cp tools/isa_guard.py /tmp/x/tools/
cp tools/scratch/ada_igemm_head.hip /tmp/x/amodal-depth-anything_amd/csrc/ada_igemm.hip; cp tools/scratch/ada_hip_head.h /tmp/x/include/ada_hip.h; cp tools/scratch/build_head.py /tmp/x/amodal-depth-anything_amd/csrc/build.py
rm -f /tmp/x/amodal-depth-anything_amd/csrc/*.so /tmp/x/amodal-depth-anything_amd/csrc/*.stamp
python /tmp/x/amodal-depth-anything_amd/csrc/build.py > $O/build_head.log 2>&1; OLD=/tmp/x/amodal-depth-anything_amd/csrc/libada_hip.so; ls -la $OLD
for i in 1 2 3; do
  for lib in $OLD ""; do
    ADA_HIP_LIB=$lib ADA_SKIP_ABI=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 1 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-new (k-walk state in registers)}', round(l['value'],1), 'img/s', round(l['ms_per_step'],2), 'ms', 'igemm frac exec', round(l['roofline']['frac_executed'],4))"
  done
done 2>&1 | tee $O/kwalk_scratch_ab.txt
