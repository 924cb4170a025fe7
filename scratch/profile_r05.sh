#!/bin/bash
# Round-5 profiles (run on the GPU box through gpurun): kernel traces and HBM-traffic PMC passes of the two legs of the bench step.
# usage: bash scratch/profile_r05.sh   -> gpurun_out/prof/r05_*.txt|json (copy the summaries into profiles/)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
HEAD=${LVT_GIT_HEAD:-unknown}      # (the box has no .git: pass `git rev-parse --short HEAD` in through the environment)
for wl in vqvae dsfvt; do
  CMD="python scratch/bench_leg.py $wl 8 3"
  rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_$wl -- $CMD > /tmp/kt_$wl.log 2>&1
  python scratch/prof_summary.py $(find /tmp/kt_$wl -name "*.db" | head -1) $OUT/r05_${wl}_kernel_stats.txt \
    "rocprofv3 --kernel-trace -- $CMD ($wl train step of bench.py x (3 warm-up + 8); round 5, git $HEAD, LVT_MATH=${LVT_MATH:-f16x2})" > /dev/null
  CMD="python scratch/bench_leg.py $wl 3 1"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/pf_$wl -- $CMD > /tmp/pf_$wl.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/pw_$wl -- $CMD > /tmp/pw_$wl.log 2>&1
  python scratch/pmc_summary.py $(find /tmp/pf_$wl -name "*.db" | head -1) $(find /tmp/pw_$wl -name "*.db" | head -1) \
    $OUT/r05_${wl}_pmc_hbm_traffic.txt $OUT/r05_${wl}_pmc_hbm_traffic.json 5 \
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- $CMD ($wl train step; 1 + 3 + 1 steps; round 5, git $HEAD, LVT_MATH=${LVT_MATH:-f16x2})" > /dev/null
done
ls -la $OUT/r05_*
head -32 $OUT/r05_vqvae_kernel_stats.txt | cut -c1-70,97-170
head -40 $OUT/r05_dsfvt_kernel_stats.txt | cut -c1-70,97-170
python - <<PY
import json
v=json.load(open("$OUT/r05_vqvae_pmc_hbm_traffic.json")); d=json.load(open("$OUT/r05_dsfvt_pmc_hbm_traffic.json"))
# one bench step = 2 VQ-VAE train steps + 1 DSFVT train step; both passes ran 1 + 3 steps
vb=(2*v["fetch_KiB_raw"]+v["write_KiB"])*1024/v["steps"]; db=(2*d["fetch_KiB_raw"]+d["write_KiB"])*1024/d["steps"]
vl=v["engine_launches"]/v["steps"]; dl=d["engine_launches"]/d["steps"]
import re
calls={wl: int(re.search(r"ENGINE_CALLS_PER_STEP (\d+)", open("/tmp/pf_%s.log" % wl).read()).group(1)) for wl in ("vqvae","dsfvt")}
for wl, dd in (("vqvae", v), ("dsfvt", d)):
    dd["engine_calls_per_step"] = calls[wl]; dd["git_head"] = "$HEAD"
    json.dump(dd, open("$OUT/r05_%s_pmc_hbm_traffic.json" % wl, "w"), indent=1)
json.dump({"hbm_bytes_per_launch": (2*vb+db)/(2*vl+dl), "engine_launches_per_step": 2*vl+dl, "hbm_bytes_per_step": 2*vb+db,
           "engine_calls_per_step": 2*calls["vqvae"]+calls["dsfvt"],
           "git_head": "$HEAD",
           "note": "combined bench step = 2 x r05_vqvae_pmc_hbm_traffic.json + 1 x r05_dsfvt_pmc_hbm_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes)"},
          open("$OUT/r05_combined_pmc_hbm_traffic.json","w"), indent=1)
print(open("$OUT/r05_combined_pmc_hbm_traffic.json").read())
PY

# instruction mix of the DSFVT leg (SQ counters, their own pass): VALU and MFMA instructions per dispatch
CMD="python scratch/bench_leg.py dsfvt 2 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format rocpd -d /tmp/sq_dsfvt -- $CMD > /tmp/sq_dsfvt.log 2>&1
python scratch/pmc_sq_summary.py $(find /tmp/sq_dsfvt -name "*.db" | head -1) $OUT/r05_dsfvt_pmc_sq.txt \
  "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -- $CMD (round 5, git $HEAD)" | head -30
