import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
rows = list(cur.execute("select name, start, end, %s from kernels where name like '%%fmx::%%' and name not like '%%probe%%' order by start" % qcol))
# take the last 60 kernels before the end
rows = rows[-int(sys.argv[2]):]
t0 = rows[0][1]
for name, s, e, q in rows:
    short = name.split("(")[0].replace("void ", "").replace("fmx::", "")[:22]
    print("%-22s q=%-4s start %8.1f  dur %7.1f  end %8.1f" % (short, q, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
