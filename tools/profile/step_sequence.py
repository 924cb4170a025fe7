import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name,start,end from kernels order by start").fetchall()
n=len(rows)
# find the first amax_multi in the second half, print following 60 kernels
for i in range(n//2, n):
    if 'amax_multi' in rows[i][0]:
        for nm,a,b in rows[i:i+70]:
            print("%-60s %8.1f" % (re.sub(r"\(.*","",nm)[:60], (b-a)/1e3))
        break
