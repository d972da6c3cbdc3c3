#!/bin/bash
# Tuning aid: build ppst_amd/libppst_hip_<name>.so with extra flags on conv_wino.hip only (timing ablations: -DWINO_ABL_*).
#   tests/build_wino_variant.sh nob -DWINO_ABL_NOB
set -e
name=$1; shift
cd "$(dirname "$0")/.."
obj=ppst_amd/csrc/_obj/conv_wino_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -fno-slp-vectorize "$@" -c ppst_amd/csrc/conv_wino.hip -o $obj
others=$(ls ppst_amd/csrc/_obj/*.o | grep -v "conv_wino\|conv_mfma_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ppst_amd/libppst_hip_$name.so $obj $others
echo built ppst_amd/libppst_hip_$name.so
