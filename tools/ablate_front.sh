#!/bin/bash
# Where does fmx::front_kernel spend its LDS cycles / time?  Diagnostic builds with phases compiled out
# (-DFMX_ABL=bits, built into sdr-j-fm_amd/lib/ab/libfmx_abl<bits>.so; results are WRONG by construction) are run
# through the bench with one PMC pass each.  On the GPU box: tools/ablate_front.sh [channels]
R=$GRAFT_REPO_ROOT; CH=${1:-512}
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 7; do
  lib=$R/sdr-j-fm_amd/lib/ab/libfmx_abl$v.so; [ $v = 0 ] && lib=$R/sdr-j-fm_amd/lib/libfmx.so
  [ -f $lib ] || continue
  FMX_LIB=$lib python $R/bench.py --channels $CH --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('abl=$v front ms', j['kernels_ms_per_step']['front_fir'], 'step ms', j['ms_per_step'])"
  FMX_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d /tmp/abl$v -o p -- python $R/bench.py --channels $CH --steps 2 --warmup 1 --no-cpu-baseline > /tmp/abl$v.log 2>&1
  python - <<PY
import sqlite3, glob
for db in glob.glob("/tmp/abl$v/p_results.db"):
    c = sqlite3.connect(db).cursor()
    rows = list(c.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%front_kernel%' group by counter_name"))
    print("   ", "  ".join("%s=%.3g" % (r[0].replace("SQ_", ""), r[1]) for r in rows))
PY
done
