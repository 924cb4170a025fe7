#!/bin/bash
# HBM traffic of the decode attention kernel (eager launches: rocprofv3 --pmc crashes on the hipGraph replay of the generation run)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof; mkdir -p $OUT
python scratch/bench_decode_attn.py > /tmp/dec.log 2>&1; cat /tmp/dec.log | tail -6
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/pf_dec -- python scratch/bench_decode_attn.py > /tmp/pf_dec.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/pw_dec -- python scratch/bench_decode_attn.py > /tmp/pw_dec.log 2>&1
python - <<PY
import sqlite3, glob, json
def q(db, c):
    cur = sqlite3.connect(glob.glob(db + "/**/*.db", recursive=True)[0]).cursor()
    # the last configuration of the script: B = 256, qi = 255 (88 launches: 8 warm + 80)
    rows = cur.execute("select counter_value, duration from pmc_events where counter_name=? and name like '%lvt_attn_decode_kernel%' order by start", (c,)).fetchall()
    rows = rows[-80:]
    return len(rows), sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows)
f, w = q("/tmp/pf_dec", "FETCH_SIZE"), q("/tmp/pw_dec", "WRITE_SIZE")
alg = 256 * 256 * 1024 * 4 * 2
d = {"kernel": "lvt_attn_decode_kernel", "config": "B = 256 videos, query position 255 (all 256 keys), 8 heads x 128, eager launches over 8 distinct caches",
     "launches": f[0], "fetch_KiB_raw": f[1], "write_KiB": w[1], "avg_us_under_pmc": f[2] / 1e3,
     "hbm_bytes_per_launch": (2 * f[1] + w[1]) * 1024, "algorithmic_kv_bytes_per_launch": alg, "git_head": "${LVT_GIT_HEAD:-unknown}",
     "timing_without_counters": open("/tmp/dec.log").read().strip().splitlines()[-3:],
     "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); separate rocprofv3 --pmc passes; the "
             "generation run itself replays hipGraphs, under which rocprofv3 --pmc segfaults (ROCm 7.0.2 on this pool)"}
json.dump(d, open("$OUT/r05_decode_attn_pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
