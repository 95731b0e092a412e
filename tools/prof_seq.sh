#!/bin/bash
# per-dispatch durations (us) of one kernel in dispatch order over a quick bench run: tools/prof_seq.sh <kernel substring> [bench args...]
R=$GRAFT_REPO_ROOT; KN=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace -d /tmp/prof_seq -o seq -- python $R/bench.py --quick "$@" > /tmp/prof_seq.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/prof_seq/**/seq_results.db", recursive=True)[0]
c = sqlite3.connect(db).cursor()
rows = list(c.execute("select name, start, end from kernels where name like '%$KN%' order by start"))
print("$KN:", " ".join("%.0f" % ((e - s) / 1e3) for n, s, e in rows))
PY
