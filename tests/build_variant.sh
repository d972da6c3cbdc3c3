#!/bin/bash
# Tuning aid: build ppst_amd/libppst_hip_<name>.so with extra -D flags on conv_mfma.hip, conv_mfma2.hip and conv_ksplit.hip.
#   tests/build_variant.sh noa -DPPST_ABL_NOA
set -e
name=$1; shift
cd "$(dirname "$0")/.."
obj=ppst_amd/csrc/_obj/conv_mfma_$name.o
obj2=ppst_amd/csrc/_obj/conv_mfma_2$name.o
obj3=ppst_amd/csrc/_obj/conv_mfma_3$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c ppst_amd/csrc/conv_mfma.hip -o $obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c ppst_amd/csrc/conv_mfma2.hip -o $obj2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c ppst_amd/csrc/conv_ksplit.hip -o $obj3
others=$(ls ppst_amd/csrc/_obj/*.o | grep -v "conv_mfma_\|conv_mfma\.o\|conv_mfma2\.o\|conv_ksplit\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ppst_amd/libppst_hip_$name.so $obj $obj2 $obj3 $others
echo built ppst_amd/libppst_hip_$name.so
