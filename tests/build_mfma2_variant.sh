#!/bin/bash
# Tuning aid: build ppst_amd/libppst_hip_<name>.so with extra flags on conv_mfma2.hip only (timing ablations of the UP9 form).
#   tests/build_mfma2_variant.sh noepi -DUP9_ABL_NOEPI
set -e
name=$1; shift
cd "$(dirname "$0")/.."
obj=ppst_amd/csrc/_obj/conv_mfma2_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c ppst_amd/csrc/conv_mfma2.hip -o $obj
others=$(ls ppst_amd/csrc/_obj/*.o | grep -v "conv_mfma2\|conv_wino_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ppst_amd/libppst_hip_$name.so $obj $others
echo built ppst_amd/libppst_hip_$name.so
