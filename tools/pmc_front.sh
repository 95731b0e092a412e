#!/bin/bash
# PMC passes over one kernel (run on the GPU box through gpurun): tools/pmc_front.sh <outdir-name> [channels] [kernel substring] [bench args]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; CH=${2:-512}; KN=${3:-front_kernel}; BARGS=${4:---steps 2 --warmup 1}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p$i -- python $R/bench.py --channels $CH $BARGS --no-cpu-baseline > $OUT.p$i.log 2>&1
done
cd $R
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob("$OUT/p*/p*_results.db")):
    c = sqlite3.connect(db).cursor()
    try:
        rows = list(c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%$KN%' group by counter_name"))
    except Exception as e:
        print(db, "ERR", e); continue
    for r in rows: print("%-28s %16.1f  (n=%d)" % r)
PY
rm -rf $OUT
