#!/bin/bash
# generation (config 5): kernel mix of the steady graph replay + HBM traffic of the decode attention kernel (gpurun)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
HEAD=${LVT_GIT_HEAD:-unknown}
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/kt_gen -- python scratch/gen_trace.py 768 > /tmp/kt_gen.log 2>&1
{ echo "# rocprofv3 --kernel-trace -- python scratch/gen_trace.py 768   (generation of 768 videos = 3 concurrent groups of 256; steady-state window of 4000 kernels;"
  echo "# kernel times overlap across the three streams; scratch/gen_gaps.py; round 5, git $HEAD)"
  python scratch/gen_gaps.py $(find /tmp/kt_gen -name "*.db" | head -1); } > $OUT/r05_generation_kernel_mix.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/pf_gen -- python scratch/gen_trace.py 256 > /tmp/pf_gen.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/pw_gen -- python scratch/gen_trace.py 256 > /tmp/pw_gen.log 2>&1
python - <<PY
import sqlite3, glob, json
def q(db, c):
    cur = sqlite3.connect(glob.glob(db + "/**/*.db", recursive=True)[0]).cursor()
    return cur.execute("select count(*), avg(counter_value), avg(duration) from pmc_events where counter_name=? and name like '%lvt_attn_decode_kernel%'", (c,)).fetchone()
f, w = q("/tmp/pf_gen", "FETCH_SIZE"), q("/tmp/pw_gen", "WRITE_SIZE")
d = {"kernel": "lvt_attn_decode_kernel", "launches": f[0], "fetch_KiB_raw": f[1], "write_KiB": w[1], "avg_us_under_pmc": f[2] / 1e3,
     "hbm_bytes_per_launch": (2 * f[1] + w[1]) * 1024, "git_head": "$HEAD",
     "note": "generation of 256 videos (one group); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); separate rocprofv3 --pmc passes"}
json.dump(d, open("$OUT/r05_generation_pmc_hbm_traffic.json", "w"), indent=1)
print(d)
PY
cat $OUT/r05_generation_kernel_mix.txt | head -14
