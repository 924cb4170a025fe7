#!/bin/bash
# kernel trace + SQ counters of the stand-alone VQ search bench (gpurun): -> gpurun_out/prof/vq_*.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
TAG=${1:-vq}
CMD="python scratch/bench_vq.py"
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_vq -- $CMD > /tmp/kt_vq.log 2>&1
tail -2 /tmp/kt_vq.log | cut -c1-150
python scratch/prof_summary.py $(find /tmp/kt_vq -name "*.db" | head -1) $OUT/${TAG}_kernel_stats.txt "rocprofv3 --kernel-trace -- $CMD" | grep lvt_vq | cut -c1-60,97-170
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format rocpd -d /tmp/sq_vq -- $CMD > /tmp/sq_vq.log 2>&1
python scratch/pmc_sq_summary.py $(find /tmp/sq_vq -name "*.db" | head -1) $OUT/${TAG}_pmc_sq.txt "rocprofv3 --pmc SQ_* -- $CMD" | grep -i "kernel\|lvt_vq\|----" | cut -c1-200
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format rocpd -d /tmp/sq2_vq -- $CMD > /tmp/sq2_vq.log 2>&1
python scratch/pmc_sq_summary.py $(find /tmp/sq2_vq -name "*.db" | head -1) $OUT/${TAG}_pmc_sq2.txt "rocprofv3 --pmc SQ_LDS_* SQ_WAIT_* -- $CMD" | grep -i "kernel\|lvt_vq\|----" | cut -c1-200
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format rocpd -d /tmp/sq3_vq -- $CMD > /tmp/sq3_vq.log 2>&1
python scratch/pmc_sq_summary.py $(find /tmp/sq3_vq -name "*.db" | head -1) $OUT/${TAG}_pmc_sq3.txt "rocprofv3 --pmc SQ_ACTIVE_* -- $CMD" | grep -i "kernel\|lvt_vq\|----" | cut -c1-200
