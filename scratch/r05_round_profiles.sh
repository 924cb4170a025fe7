#!/bin/bash
# round-5 evidence run (gpurun): kernel traces + PMC traffic of both legs, the DP overlap timeline, the generation kernel mix, bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
bash scratch/profile_r05.sh > $OUT/r05_profile_log.txt 2>&1
tail -n 45 $OUT/r05_profile_log.txt
# DP overlap: one-rank RCCL group, reducers active
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_dp -- python scratch/dp_overlap_trace.py 4 > /tmp/kt_dp.log 2>&1
tail -n 3 /tmp/kt_dp.log
python scratch/dp_overlap_summary.py $(find /tmp/kt_dp -name "*.db" | head -1) $OUT/r05_dp_overlap_timeline.txt \
  "rocprofv3 --kernel-trace -- python scratch/dp_overlap_trace.py 4 (DSFVT train steps, b = 64, ONE-rank RCCL group, reducers active: LVT_DP_SINGLE_RANK; round 5, git ${LVT_GIT_HEAD:-unknown})" | head -40
