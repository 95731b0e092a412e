#!/bin/bash
# PMC passes over several kernels: tools/pmc2.sh "<kernel substrings, space separated>" [bench args...]   (on the GPU box)
R=$GRAFT_REPO_ROOT; KNS=$1; shift
cd /tmp && export TMPDIR=/tmp
for CNT in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
rm -rf /tmp/pmck
rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmck -o p -- python $R/bench.py --quick "$@" > /tmp/pmck.log 2>&1
python - <<PY
import sqlite3, glob
for db in glob.glob("/tmp/pmck/p_results.db"):
    c = sqlite3.connect(db).cursor()
    for kn in "$KNS".split():
        rows = list(c.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%" + kn + "%' group by counter_name"))
        print(kn + ":", "  ".join("%s=%.4g" % (r[0].replace("SQ_", ""), r[1]) for r in rows))
PY
done
