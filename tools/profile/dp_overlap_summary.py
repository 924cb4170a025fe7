"""Timeline summary of tools/profile/dp_overlap_trace.py from a rocprofv3 rocpd database: for every train step (delimited by the
optimizer launches) the start / end of each RCCL all-reduce kernel relative to the backward pass it overlaps.
   python tools/profile/dp_overlap_summary.py DB OUT TITLE"""
import re, sqlite3, sys
db, out, title = sys.argv[1:4]
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = cur.execute("select name, start, end%s from kernels order by start" % ((", " + sid) if sid else "")).fetchall()
def short(n): return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))[:60]
isnccl = lambda n: re.search(r"nccl|rccl|AllReduce|all_reduce", n, re.I) is not None
opt = [i for i, r in enumerate(rows) if "lvt_rmsprop" in r[0]]
# steps = runs between the LAST optimizer launch of one step and the FIRST of the next
bounds, prev = [], None
for i in opt:
    if prev is None or i - prev > 50:
        bounds.append(i)
    prev = i
lines = ["# " + title, "# columns of `kernels`: " + ", ".join(cols),
         "# per train step: backward = first lvt_xent_bwd launch .. last lvt_* launch before the optimizer; every RCCL kernel with its "
         "start / end relative to the START and to the END of that backward pass (us; negative 'to_bwd_end' = it began before "
         "the backward pass had finished = overlapped)"]
for si in range(1, len(bounds)):
    lo, hi = bounds[si - 1], bounds[si]
    seg = rows[lo:hi]
    bstart = next((r[1] for r in seg if "xent_bwd" in r[0]), None)
    comp = [r for r in seg if not isnccl(r[0]) and r[0].lstrip("void ").startswith(("lvt_", "(anonymous"))] or [r for r in seg if not isnccl(r[0])]
    bend = max(r[2] for r in comp if bstart is None or r[1] >= bstart)
    nc = [r for r in seg if isnccl(r[0])]
    if bstart is None:
        continue
    lines.append("step %d: backward %.1f us (%d compute launches after its start); %d RCCL launches; optimizer starts %.1f us after backward end"
                 % (si, (bend - bstart) / 1e3, sum(1 for r in comp if r[1] >= bstart), len(nc), (rows[hi][1] - bend) / 1e3))
    for r in nc:
        lines.append("    %-58s start +%9.1f us  dur %7.1f us  start-to-bwd-end %9.1f us  end-to-bwd-end %9.1f us%s"
                     % (short(r[0]), (r[1] - bstart) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - bend) / 1e3, (r[2] - bend) / 1e3,
                        ("  stream %s" % r[3]) if sid else ""))
    last = max((r[2] for r in nc), default=bend)
    lines.append("    -> last RCCL kernel ends %.1f us %s the last backward kernel; all but the last bucket started before backward end: %s"
                 % (abs(last - bend) / 1e3, "after" if last > bend else "before", all(r[1] < bend for r in nc[:-1]) if nc else None))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
