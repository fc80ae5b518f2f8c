#!/bin/bash
# Builds experiment variants of libxeve_hip.so that differ in walk.hip's switches only (run HERE, before gpurun: the .so files travel with the snapshot):
#   xeve_amd/lib/exp/libxeve_hip_<name>.so ; a run selects one with XEVE_HIP_LIB_PATH.  usage: tools/build_walk_variants.sh  (then: gpurun -- 'bash tools/gpu/r05_variants.sh')
set -e
cd "$(dirname "$0")/.."
make -s -C xeve_amd/csrc
mkdir -p xeve_amd/lib/exp xeve_amd/csrc/build_exp
OBJS=$(ls xeve_amd/csrc/build/*.o | grep -v "/walk.o")
build() { # name, flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $2 -c xeve_amd/csrc/walk.hip -o xeve_amd/csrc/build_exp/walk_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o xeve_amd/lib/exp/libxeve_hip_$1.so $OBJS xeve_amd/csrc/build_exp/walk_$1.o -ldl
  echo "built xeve_amd/lib/exp/libxeve_hip_$1.so ($2)"
}
build stages "-DXW_NOINLINE_STAGES=1"
build pairs "-DXW_CODER_PAIRS=1"
build stages_pairs "-DXW_NOINLINE_STAGES=1 -DXW_CODER_PAIRS=1"
build wg5 "-DXW_WG_PER_CU=5"
