"""Kernel names before / after every launch whose name matches PATTERN (rocpd kernel-trace db), counted by context."""
import collections, re, sqlite3, sys
db, pat = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("void ", "")[:60]
ctx = collections.Counter()
for i, (n, s, e) in enumerate(rows):
    if re.search(pat, n):
        prev = short(rows[i - 1][0]) if i else "-"
        nxt = short(rows[i + 1][0]) if i + 1 < len(rows) else "-"
        ctx[(prev, short(n)[:50], nxt, )] += 1
for (p, n, x), c in ctx.most_common(25):
    print("%5d  %-52s | after %-52s | before %s" % (c, n, p, x))
