#!/bin/bash
# Build ablation / experiment variants of one kernel source into tools/ubench/<dir>/<prefix>_<name>.so
#   bash tools/build_variants.sh pose2room_amd/csrc/stgcn_gcn2.hip g2 g2 base: sload:-DG2X_SLOAD
SRC=$1; DIR=$2; PRE=$3; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/ubench/$DIR
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -shared \
      -I$R/pose2room_amd/csrc $flags -o $R/tools/ubench/$DIR/${PRE}_${name}.so $R/$SRC &
done
wait
ls -la $R/tools/ubench/$DIR
