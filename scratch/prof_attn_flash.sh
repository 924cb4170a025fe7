#!/bin/bash
# kernel trace + SQ counters of the stand-alone flash attention bench (gpurun): -> gpurun_out/prof/attn_flash_*.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
TAG=${1:-attn_flash}
CMD="python scratch/bench_attn_flash.py both noplanes"
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_af -- $CMD > /tmp/kt_af.log 2>&1
python scratch/prof_summary.py $(find /tmp/kt_af -name "*.db" | head -1) $OUT/${TAG}_kernel_stats.txt "rocprofv3 --kernel-trace -- $CMD" | cut -c1-60,97-170 | head -14
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format rocpd -d /tmp/sq_af -- $CMD > /tmp/sq_af.log 2>&1
python scratch/pmc_sq_summary.py $(find /tmp/sq_af -name "*.db" | head -1) $OUT/${TAG}_pmc_sq.txt "rocprofv3 --pmc SQ_* -- $CMD" | cut -c1-64,65-200 | head -12
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format rocpd -d /tmp/sq2_af -- $CMD > /tmp/sq2_af.log 2>&1
python scratch/pmc_sq_summary.py $(find /tmp/sq2_af -name "*.db" | head -1) $OUT/${TAG}_pmc_sq2.txt "rocprofv3 --pmc SQ_LDS_* SQ_WAIT_* -- $CMD" | head -12
