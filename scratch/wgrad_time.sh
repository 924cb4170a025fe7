#!/bin/bash
# kernel times of the frame-resident weight-gradient kernels inside the VQ-VAE train step, for the library in LVT_HIP_LIB
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
rm -rf /tmp/kt_w; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_w -- python scratch/bench_leg.py vqvae 6 2 > /tmp/kt_w.log 2>&1
python scratch/prof_summary.py $(find /tmp/kt_w -name "*.db" | head -1) /tmp/kt_w.txt "wgrad" > /dev/null
grep "wgrad_frames\|TOTAL" /tmp/kt_w.txt | cut -c1-60,97-170
