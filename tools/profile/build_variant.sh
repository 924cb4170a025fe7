#!/bin/bash
# usage: tools/profile/build_variant.sh NAME "-DFLAG ..." [unit, default gemm_engine]   -> build/variants/lib_NAME.so
set -e
cd "$(dirname "$0")/../../lvt_amd/csrc"
mkdir -p ../../build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result $2 -c ${3:-gemm_engine}.hip -o ../../build/variants/ge_$1.o
OTHERS=$(ls *.o | grep -v ${3:-gemm_engine}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS ../../build/variants/ge_$1.o -o ../../build/variants/lib_$1.so
rm ../../build/variants/ge_$1.o
