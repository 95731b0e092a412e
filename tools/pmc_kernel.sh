#!/bin/bash
# one PMC pass over the kernels matching a substring: tools/pmc_kernel.sh "<counters>" <kernel substring> [bench args...]
R=$GRAFT_REPO_ROOT; CNT=$1; KN=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmck
rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmck -o p -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/pmck.log 2>&1
python - <<PY
import sqlite3, glob
for db in glob.glob("/tmp/pmck/p_results.db"):
    c = sqlite3.connect(db).cursor()
    rows = list(c.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%$KN%' group by counter_name"))
    print("$KN:", "  ".join("%s=%.4g" % (r[0].replace("SQ_", ""), r[1]) for r in rows))
PY
