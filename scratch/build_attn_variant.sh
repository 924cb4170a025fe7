#!/bin/bash
# usage: scratch/build_attn_variant.sh NAME "-DFLAG ..."   -> scratch/variants/libattn_NAME.so (attention_pipe.hip rebuilt with the flags)
set -e
cd "$(dirname "$0")/../lvt_amd/csrc"
mkdir -p ../../scratch/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result $2 -c attention_pipe.hip -o ../../scratch/variants/ap_$1.o
OTHERS=$(ls *.o | grep -v attention_pipe.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS ../../scratch/variants/ap_$1.o -o ../../scratch/variants/libattn_$1.so
rm ../../scratch/variants/ap_$1.o
