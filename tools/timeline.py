#!/usr/bin/env python3
"""Kernel timeline of one bench step from a rocprofv3 results .db: tools/timeline.py <db> [step index from the end]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute("select name, start, end, %s from kernels where name like '%%fmx::%%' order by start" % (qcol or "0")))
# split into steps at each front_kernel
starts = [i for i, r in enumerate(rows) if "front_kernel" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a = starts[-k]; b = starts[-k + 1] if k > 1 else len(rows)
t0 = rows[a][1]
print("# columns:", cols)
last_end = {}
for name, s, e, q in rows[a:b]:
    short = name.split("(")[0].replace("void ", "").replace("fmx::", "")[:22]
    print("%-22s q=%-4s start %8.1f us  dur %7.1f us  end %8.1f" % (short, q, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
