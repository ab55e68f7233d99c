#!/bin/bash
# Instrumented build of the C-ABI library for the cycle traces: every csrc/*.hip with -DP2R_CYCLE_TRACE ->
# tools/ubench/libp2r_hip_trace.so (git-ignored; travels to the GPU box).  The product library is not touched.
#   bash tools/build_trace_lib.sh && gpurun -- 'python tools/dev_t3_trace.py; python tools/dev_g3_trace.py'
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/tools/ubench/trace_obj
mkdir -p $O
for f in $R/pose2room_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -DP2R_CYCLE_TRACE \
      -I$R/pose2room_amd/csrc -c $f -o $O/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ubench/libp2r_hip_trace.so $O/*.o
ls -la $R/tools/ubench/libp2r_hip_trace.so
