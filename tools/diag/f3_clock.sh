#!/bin/bash
# effective clock and wave residency of the stage-A kernels: GRBM_GUI_ACTIVE / duration (on the GPU box): tools/diag/f3_clock.sh
cd /tmp && export TMPDIR=/tmp
for fk in 2 1; do
  rm -rf /tmp/pmcc
  FMX_FRONT_KERNEL=$fk rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_BUSY_CU_CYCLES -d /tmp/pmcc -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --quick --steps 4 --warmup 44 > /tmp/pmcc.log 2>&1
  python - <<PY
import sqlite3, glob
for db in glob.glob("/tmp/pmcc/p_results.db"):
    c = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if 'kernel_dispatch' in t and 'rocpd' in t]
    print([t for t in tabs if 'kernel' in t][:8])
    try:
        rows = list(c.execute("select name, avg(end - start), count(*) from kernels where name like '%front%' group by name"))
        print(rows)
    except Exception as e:
        print("ERR", e)
    rows = list(c.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%front%' group by kernel_name, counter_name"))
    for r in rows: print(r[0][:40], r[1], "%.4g" % r[2])
PY
done
