#!/bin/bash
# Tuning aid: build ppst_amd/libppst_hip_<name>.so with extra -D flags on conv_mfma.hip and conv_mfma2.hip.
#   tests/build_variant.sh noa -DPPST_ABL_NOA
set -e
name=$1; shift
cd "$(dirname "$0")/.."
obj=ppst_amd/csrc/_obj/conv_mfma_$name.o
obj2=ppst_amd/csrc/_obj/conv_mfma_2$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c ppst_amd/csrc/conv_mfma.hip -o $obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c ppst_amd/csrc/conv_mfma2.hip -o $obj2
others=$(ls ppst_amd/csrc/_obj/*.o | grep -v "conv_mfma_\|conv_mfma\.o\|conv_mfma2\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ppst_amd/libppst_hip_$name.so $obj $obj2 $others
echo built ppst_amd/libppst_hip_$name.so
