"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel stats table."""
import re
import sqlite3
import sys

db, out, title = sys.argv[1], sys.argv[2], sys.argv[3]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by name order by 3 desc").fetchall()
span = cur.execute("select min(start), max(end) from kernels").fetchone()
tot = sum(r[2] for r in rows)
lines = ["# " + title, "# times in microseconds; pct = share of total kernel time",
         "%-96s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
for n, c, t, a, mn, mx in rows[:45]:
    n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))[:94]
    lines.append("%-96s %7d %12.1f %10.1f %10.1f %10.1f %6.2f" % (n, c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100 * t / tot))
lines.append("TOTAL kernel time %.1f us over %d launches; first-to-last kernel span %.1f us (GPU busy %.1f%%)"
             % (tot / 1e3, sum(r[1] for r in rows), (span[1] - span[0]) / 1e3, 100 * tot / (span[1] - span[0])))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
