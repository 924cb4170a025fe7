"""Idle time between consecutive kernels of a bench leg (one stream): rocprofv3 --kernel-trace --output-format rocpd -d DIR -- python
tools/profile/bench_leg.py LEG 6 2; python tools/profile/step_gaps.py DIR/**/*.db"""
import sqlite3, sys, re, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name,start,end from kernels order by start").fetchall()
n = len(rows); w = rows[int(n * 0.5):int(n * 0.9)]
busy = sum(b - a for _, a, b in w); span = w[-1][2] - w[0][1]
gaps = [w[i + 1][1] - w[i][2] for i in range(len(w) - 1)]
print("window: %d kernels, span %.1f us, busy %.1f us (%.1f%%), avg kernel %.2f us" % (len(w), span / 1e3, busy / 1e3, 100 * busy / span, busy / len(w) / 1e3))
gs = sorted(gaps)
print("gap between a kernel's end and the next one's start: mean %.2f us, median %.2f, p10 %.2f, p90 %.2f, total %.1f us (%.1f%% of the span)" % (
    sum(gaps) / len(gaps) / 1e3, gs[len(gs) // 2] / 1e3, gs[len(gs) // 10] / 1e3, gs[9 * len(gs) // 10] / 1e3, sum(gaps) / 1e3, 100 * sum(gaps) / span))
by = collections.defaultdict(lambda: [0, 0])
for i, g in enumerate(gaps):
    k = re.sub(r"\(.*", "", w[i + 1][0])[:56]; by[k][0] += 1; by[k][1] += g
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  before %-58s %5d x %6.2f us" % (k, c, t / c / 1e3))
if len(sys.argv) > 2:      # the neighbourhood of every gap above N us
    thr = float(sys.argv[2]) * 1e3
    for i, g in enumerate(gaps):
        if g > thr:
            print("gap %7.1f us: %s [%.1f us]  ->  %s [%.1f us] -> %s" % (g / 1e3, re.sub(r"\(.*", "", w[i][0])[:50], (w[i][2] - w[i][1]) / 1e3,
                  re.sub(r"\(.*", "", w[i + 1][0])[:50], (w[i + 1][2] - w[i + 1][1]) / 1e3, re.sub(r"\(.*", "", w[i + 2][0])[:40] if i + 2 < len(w) else ""))
