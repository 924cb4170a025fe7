#!/bin/bash
# usage: tools/profile/kt.sh TAG leg steps   -> gpurun_out/kt_TAG.txt  (kernel-trace summary of tools/profile/bench_leg.py; env passes through)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/kt_$1
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_$1 -- python tools/profile/bench_leg.py $2 ${3:-6} 2 > /tmp/kt_$1.log 2>&1
tail -n 2 /tmp/kt_$1.log
python tools/profile/prof_summary.py $(find /tmp/kt_$1 -name "*.db" | head -1) gpurun_out/kt_$1.txt "kernel trace $2 x $3 ($1)" > /dev/null
head -24 gpurun_out/kt_$1.txt
