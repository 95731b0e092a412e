#!/usr/bin/env python3
"""Kernel timeline of a window of a rocprofv3 results .db: tools/timeline_window.py <db> <window ms> [offset from the last fmx kernel, ms]   (diagnostic)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(cur.execute("select name, start, end, %s from kernels where name like '%%fmx::%%' and name not like '%%probe%%' order by start" % qcol))
if len(sys.argv) > 4:      # everything the GPU ran: other kernels and memory copies too
    allk = list(cur.execute("select name, start, end, %s from kernels where name not like '%%fmx::%%' order by start" % qcol))
    try: allk += [("COPY " + str(r[0]), r[1], r[2], -1) for r in cur.execute("select name, start, end from memory_copies")]
    except Exception as e: print("# no memory copies:", e)
    last = rows[-1][2]
    rows = sorted(rows + allk, key=lambda r: r[1]); rows = [r for r in rows if r[1] <= last]
win = float(sys.argv[2]) * 1e6; off = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0.0
t1 = rows[-1][2] - off; t0 = t1 - win
for name, s, e, q in rows:
    if s < t0 or s > t1: continue
    short = name.split("(")[0].replace("void ", "").replace("fmx::", "")[:26]
    print("%-26s q=%-4s start %9.1f us  dur %7.1f us  end %9.1f" % (short, q, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
