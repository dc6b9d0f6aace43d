#!/bin/bash
# round 4, session z: what the memory path charges for the shape of a wave's store (the measurement behind r04_n)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4zz
O=$PWD/gpurun_out/r4zz
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/store_segments tools/ubench/store_segments.hip 2>/dev/null && timeout 300 /tmp/store_segments 20 > $O/store_segments.txt 2>&1
cat $O/store_segments.txt
