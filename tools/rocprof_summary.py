#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (rocpd sqlite) into the text table committed under profiles/.

usage: tools/rocprof_summary.py gpurun_out/prof_xxx/name_results.db [bench.log] > profiles/rNN_xxx.txt
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
    if len(sys.argv) > 2:
        for line in open(sys.argv[2]):
            if line.startswith("{"):
                print("# bench line: " + line.strip())
    print("%-58s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0][-58:]
        print("%-58s %6d %12.1f %12.2f %7.2f" % (short, calls, total / 1.0, avg / 1.0, pct))
    print()
    print("# per-dispatch resources of the fmx kernels (first dispatch of each)")
    seen = set()
    q = ("select name,grid_x,grid_y,workgroup_x,lds_size,static_lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count,duration "
         "from kernels where name like 'fmx::%' or name like 'void fmx::%' order by start")
    rows = list(cur.execute(q))
    for r in rows:
        k = r[0].split("(")[0]
        if k in seen:
            continue
        seen.add(k)
        print("%-24s grid=(%d,%d) wg=%d lds=%d static_lds=%d scratch=%d vgpr=%d agpr=%d sgpr=%d" %
              (k, r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]))
    print()
    print("# per-kernel duration distribution (ns) of the fmx kernels")
    for k in sorted(seen):
        d = sorted(r[10] for r in rows if r[0].split("(")[0] == k)
        n = len(d)
        print("%-24s n=%d min=%d median=%d max=%d mean=%.0f" % (k, n, d[0], d[n // 2], d[-1], sum(d) / n))
    try:
        rows = list(cur.execute("select * from counters_collection limit 1"))
        if rows:
            print()
            print("# PMC counters (sum over dispatches, per kernel)")
            cols = [d[0] for d in cur.description]
            ni, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else None, None, None
            q = "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"
            for kn, cn, v, c in cur.execute(q):
                if kn.startswith("fmx::"):
                    print("%-24s %-28s sum=%.6g dispatches=%d per_dispatch=%.6g" % (kn.split("(")[0], cn, v, c, v / c))
    except sqlite3.Error as e:
        print("# (no counters: %s)" % e)


if __name__ == "__main__":
    main()
