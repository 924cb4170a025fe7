#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): kernel traces and HBM-traffic PMC passes of the bench command.
# usage: bash scratch/profile_r02.sh   -> gpurun_out/prof/*.txt|json (copy the summaries into profiles/)
set -x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
VQ="python bench.py --steps 10 --warmup 3 --no-dsfvt --no-generate --no-cpu-baseline --no-strict-f32"
DS="python scratch/bench_leg.py dsfvt"
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt_vq -- $VQ > $OUT/kt_vq.log 2>&1
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/kt_ds -- $DS > $OUT/kt_ds.log 2>&1
python scratch/prof_summary.py $(find $OUT/kt_vq -name "*.db" | head -1) $OUT/r02_vqvae_kernel_stats.txt \
  "rocprofv3 --kernel-trace -- $VQ (VQ-VAE train step x (3 + 10 + 10 instrumented); round 2)"
python scratch/prof_summary.py $(find $OUT/kt_ds -name "*.db" | head -1) $OUT/r02_dsfvt_kernel_stats.txt \
  "rocprofv3 --kernel-trace -- $DS (DSFVT train step x (2 + 6 + 6 instrumented), 64 slices; round 2)"
for wl in vq ds; do
  if [ $wl = vq ]; then CMD="python bench.py --steps 3 --warmup 1 --no-dsfvt --no-generate --no-cpu-baseline --no-strict-f32"; else CMD="python scratch/bench_leg.py dsfvt 2 1"; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $OUT/pmc_f_$wl -- $CMD > $OUT/pmc_f_$wl.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $OUT/pmc_w_$wl -- $CMD > $OUT/pmc_w_$wl.log 2>&1
done
python scratch/pmc_summary.py $(find $OUT/pmc_f_vq -name "*.db" | head -1) $(find $OUT/pmc_w_vq -name "*.db" | head -1) \
  $OUT/r02_vqvae_pmc_hbm_traffic.txt $OUT/r02_vqvae_pmc_hbm_traffic.json 7 \
  "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- bench.py --steps 3 --warmup 1 (VQ-VAE train step, 512 frames; 1 + 3 + 3 instrumented steps; round 2)"
python scratch/pmc_summary.py $(find $OUT/pmc_f_ds -name "*.db" | head -1) $(find $OUT/pmc_w_ds -name "*.db" | head -1) \
  $OUT/r02_dsfvt_pmc_hbm_traffic.txt $OUT/r02_dsfvt_pmc_hbm_traffic.json 5 \
  "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- scratch/bench_leg.py dsfvt 2 1 (DSFVT train step, 64 slices; 1 + 2 + 2 instrumented steps; round 2)"
ls -la $OUT/*.txt $OUT/*.json
