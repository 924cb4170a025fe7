#!/bin/bash
# kernel time (rocprofv3 kernel trace) of vq.nearest with the product library and every variant scratch/variants/lib_vq_*.so
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in product $(ls scratch/variants/lib_vq_*.so 2>/dev/null); do
  if [ "$lib" = product ]; then unset LVT_HIP_LIB; else export LVT_HIP_LIB=$PWD/$lib; fi
  rm -rf /tmp/kt_v; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_v -- python scratch/bench_vq.py > /tmp/kt_v.log 2>&1
  echo "== $lib"; tail -2 /tmp/kt_v.log | cut -c1-70; python scratch/prof_summary.py $(find /tmp/kt_v -name "*.db" | head -1) /tmp/kt_v.txt x | grep "lvt_vq" | cut -c1-56,97-150
done
