import sqlite3, re, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name,start,end from kernels order by start").fetchall()
# take a window in the last third of the run (steady graph replay)
n = len(rows); w = rows[int(n * 0.7):int(n * 0.7) + 4000]
busy = sum(b - a for _, a, b in w); span = w[-1][2] - w[0][1]
print("window: %d kernels, span %.1f us, busy %.1f us (%.1f%%), avg kernel %.2f us, avg gap %.2f us" % (
    len(w), span / 1e3, busy / 1e3, 100 * busy / span, busy / len(w) / 1e3, (span - busy) / len(w) / 1e3))
import collections
c = collections.defaultdict(lambda: [0, 0])
for nme, a, b in w:
    k = re.sub(r"\(.*", "", nme)[:60]; c[k][0] += 1; c[k][1] += b - a
for k, (cnt, t) in sorted(c.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-62s %6d %9.1f us total %7.2f us avg" % (k, cnt, t / 1e3, t / cnt / 1e3))
