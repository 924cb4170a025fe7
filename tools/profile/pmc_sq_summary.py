"""Summarise a rocprofv3 --pmc pass of SQ / GRBM counters per kernel (engine kernels only)."""
import re, sqlite3, sys
db, out, title = sys.argv[1:4]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events group by name, counter_name").fetchall()
res = {}
for n, c, cnt, avg, dur in rows:
    k = re.sub(r"\((?!anonymous).*", "", n.replace("(anonymous namespace)::", ""))[:64]
    res.setdefault(k, {"calls": cnt, "us": dur / 1e3})[c] = avg
names = sorted({c for v in res.values() for c in v if c not in ("calls", "us")})
lines = ["# " + title, "# per-dispatch averages", "%-56s %6s %9s " % ("kernel", "calls", "avg_us") + " ".join("%22s" % c for c in names)]
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["us"] * kv[1]["calls"])[:12]:
    lines.append("%-56s %6d %9.1f " % (k, v["calls"], v["us"]) + " ".join("%22.4g" % v.get(c, float("nan")) for c in names))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
