#!/bin/bash
# round 4, session a: new tests (graph/LRU, tail stress, torchrun preflight, new fixtures), packed-fp32 microbenchmark, batch sweep, group sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
O=$PWD/gpurun_out/r4a
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_ubench tools/ubench/pk_f32_beside_mfma.hip 2>/dev/null && timeout 300 /tmp/pk_ubench 4000 5 > $O/pk_f32_beside_mfma.txt 2>&1; cat $O/pk_f32_beside_mfma.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q -s -x -k "golden or lru or preflight or stress or graph" 2>&1 | grep -v amdgpu | tail -n 60 > $O/pytest_new.txt; grep "rel-L1\|passed\|failed\|stress\|torchrun" $O/pytest_new.txt
timeout 600 python tools/batch_sweep.py 2>&1 | grep -v amdgpu > $O/batch_sweep.txt; cat $O/batch_sweep.txt
timeout 600 python tools/group_sweep.py 2>&1 | grep -v amdgpu > $O/group_sweep.txt; cat $O/group_sweep.txt
