import re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events group by name, counter_name").fetchall()
for n, c, cnt, avg, dur in rows:
    if "lvt_gemm_kernel" in n: print("%-30s %-28s n=%d avg=%.5g dur_us=%.1f" % (re.sub(r"\(.*", "", n)[5:35], c, cnt, avg, dur/1e3))
