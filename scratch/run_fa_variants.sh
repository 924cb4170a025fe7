#!/bin/bash
# time the backward attention kernels of every variant library under scratch/variants (rocprofv3 kernel trace, B kernel rows)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in product $(ls scratch/variants/lib_fa_*.so 2>/dev/null); do
  if [ "$lib" = product ]; then unset LVT_HIP_LIB; else export LVT_HIP_LIB=$PWD/$lib; fi
  rm -rf /tmp/kt_v; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_v -- python scratch/bench_attn_flash.py bwd noplanes > /tmp/kt_v.log 2>&1
  echo "== $lib"; python scratch/prof_summary.py $(find /tmp/kt_v -name "*.db" | head -1) /tmp/kt_v.txt x | grep "lvt_attn_bwd" | cut -c1-56,97-150
done
