#!/bin/bash
# Evidence run of a round (on the GPU box, through gpurun): kernel traces + HBM-traffic PMC passes of the two legs of the bench
# step, SQ instruction counters of the DSFVT leg (default path and LVT_P2=1), the DP overlap timeline, the generation kernel mix.
# usage: LVT_GIT_HEAD=$(git rev-parse --short HEAD) ROUND=r06 bash tools/profile/round_profiles.sh  -> gpurun_out/prof/${ROUND}_*
# (copy the summaries into profiles/).  Counter passes are their own runs with --kernel-trace only.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=${ROUND:-r06}
OUT=gpurun_out/prof; mkdir -p $OUT
HEAD=${LVT_GIT_HEAD:-unknown}      # (the box has no .git: pass `git rev-parse --short HEAD` in through the environment)
for wl in vqvae dsfvt; do
  CMD="python tools/profile/bench_leg.py $wl 8 3"
  rm -rf /tmp/kt_$wl /tmp/pf_$wl /tmp/pw_$wl
  rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_$wl -- $CMD > /tmp/kt_$wl.log 2>&1
  python tools/profile/prof_summary.py $(find /tmp/kt_$wl -name "*.db" | head -1) $OUT/${R}_${wl}_kernel_stats.txt \
    "rocprofv3 --kernel-trace -- $CMD ($wl train step of bench.py x (3 warm-up + 8); $R, git $HEAD, LVT_MATH=${LVT_MATH:-f16x2})" > /dev/null
  CMD="python tools/profile/bench_leg.py $wl 3 1"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/pf_$wl -- $CMD > /tmp/pf_$wl.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/pw_$wl -- $CMD > /tmp/pw_$wl.log 2>&1
  python tools/profile/pmc_summary.py $(find /tmp/pf_$wl -name "*.db" | head -1) $(find /tmp/pw_$wl -name "*.db" | head -1) \
    $OUT/${R}_${wl}_pmc_hbm_traffic.txt $OUT/${R}_${wl}_pmc_hbm_traffic.json 5 \
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- $CMD ($wl train step; 1 + 3 + 1 steps; $R, git $HEAD, LVT_MATH=${LVT_MATH:-f16x2})" > /dev/null
done
python - <<PY
import json, re
R, OUT, HEAD = "$R", "$OUT", "$HEAD"
v=json.load(open("%s/%s_vqvae_pmc_hbm_traffic.json" % (OUT, R))); d=json.load(open("%s/%s_dsfvt_pmc_hbm_traffic.json" % (OUT, R)))
# one bench step = 2 VQ-VAE train steps + 1 DSFVT train step; both passes ran 1 + 3 + 1 steps
vb=(2*v["fetch_KiB_raw"]+v["write_KiB"])*1024/v["steps"]; db=(2*d["fetch_KiB_raw"]+d["write_KiB"])*1024/d["steps"]
vl=v["engine_launches"]/v["steps"]; dl=d["engine_launches"]/d["steps"]
calls={wl: int(re.search(r"ENGINE_CALLS_PER_STEP (\d+)", open("/tmp/pf_%s.log" % wl).read()).group(1)) for wl in ("vqvae","dsfvt")}
for wl, dd in (("vqvae", v), ("dsfvt", d)):
    dd["engine_calls_per_step"] = calls[wl]; dd["git_head"] = HEAD
    json.dump(dd, open("%s/%s_%s_pmc_hbm_traffic.json" % (OUT, R, wl), "w"), indent=1)
json.dump({"hbm_bytes_per_launch": (2*vb+db)/(2*vl+dl), "engine_launches_per_step": 2*vl+dl, "hbm_bytes_per_step": 2*vb+db,
           "engine_calls_per_step": 2*calls["vqvae"]+calls["dsfvt"], "git_head": HEAD,
           "note": "combined bench step = 2 x %s_vqvae_pmc_hbm_traffic.json + 1 x %s_dsfvt_pmc_hbm_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes)" % (R, R)},
          open("%s/%s_combined_pmc_hbm_traffic.json" % (OUT, R),"w"), indent=1)
print(open("%s/%s_combined_pmc_hbm_traffic.json" % (OUT, R)).read())
PY
# instruction mix of the DSFVT leg (SQ counters, their own passes): default path, and with the plane-fed products (LVT_P2=1)
CMD="python tools/profile/bench_leg.py dsfvt 2 1"
for tag in default p2; do
  rm -rf /tmp/sq_$tag
  if [ $tag = p2 ]; then export LVT_P2=1; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format rocpd -d /tmp/sq_$tag -- $CMD > /tmp/sq_$tag.log 2>&1
  python tools/profile/pmc_sq_summary.py $(find /tmp/sq_$tag -name "*.db" | head -1) $OUT/${R}_dsfvt_pmc_sq_$tag.txt \
    "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -- $CMD ($R, git $HEAD, $tag: LVT_P2=${LVT_P2:-unset})" | head -16
  rm -rf /tmp/sw_$tag
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format rocpd -d /tmp/sw_$tag -- $CMD > /tmp/sw_$tag.log 2>&1
  python tools/profile/pmc_sq_summary.py $(find /tmp/sw_$tag -name "*.db" | head -1) $OUT/${R}_dsfvt_pmc_sq_wait_$tag.txt \
    "rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -- $CMD ($R, git $HEAD, $tag)" | head -16
done
unset LVT_P2
# DP overlap: one-rank RCCL group, reducers active (bucket order measured in the first backward)
python tools/profile/dp_overlap_trace.py 4 > $OUT/${R}_dp_overlap_timeline.txt 2>/tmp/dp.err; tail -n 30 $OUT/${R}_dp_overlap_timeline.txt
# generation kernel mix
rm -rf /tmp/kt_gen
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_gen -- python tools/profile/gen_trace.py 768 > /tmp/kt_gen.log 2>&1
{ echo "# rocprofv3 --kernel-trace -- python tools/profile/gen_trace.py 768   (generation of 768 videos = 3 concurrent groups of 256; steady-state window of 4000 kernels;"
  echo "# kernel times overlap across the three streams; tools/profile/gen_gaps.py; $R, git $HEAD)"
  python tools/profile/gen_gaps.py $(find /tmp/kt_gen -name "*.db" | head -1); } > $OUT/${R}_generation_kernel_mix.txt
head -14 $OUT/${R}_generation_kernel_mix.txt
ls -la $OUT/${R}_*
