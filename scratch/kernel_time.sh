#!/bin/bash
# usage: scratch/kernel_time.sh vqvae|dsfvt PATTERN  -> per-kernel times (rocprofv3 kernel trace of 2 + 6 steps of that leg) matching PATTERN
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rm -rf /tmp/kt_x; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_x -- python scratch/bench_leg.py $1 6 2 > /tmp/kt_x.log 2>&1
python scratch/prof_summary.py $(find /tmp/kt_x -name "*.db" | head -1) /tmp/kt_x.txt "$1" > /dev/null
grep -E "$2|TOTAL" /tmp/kt_x.txt | cut -c1-60,97-170
