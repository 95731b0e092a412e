R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 6 7; do
  lib=$R/sdr-j-fm_amd/lib/ab/libfmx_abl$v.so; [ $v = 0 ] && lib=$R/sdr-j-fm_amd/lib/libfmx.so
  rm -rf /tmp/abl$v
  FMX_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d /tmp/abl$v -o p -- python $R/bench.py --quick --no-cpu-baseline > /tmp/abl$v.log 2>&1
  python - <<PY
import sqlite3, glob, json
ms = json.loads([l for l in open("/tmp/abl$v.log") if l.startswith("{")][-1])["kernels_ms_per_step"]["front_fir"]
for db in glob.glob("/tmp/abl$v/p_results.db"):
    c = sqlite3.connect(db).cursor()
    rows = dict(c.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%front_kernel%' group by counter_name"))
    tiles = 4096 * 150.0
    print("ABL=$v front %.3f ms | per tile: VALU %.0f  LDS %.0f  SALU %.0f | LDS active %.0f conflict %.0f cycles per tile-CU... " % (ms, rows.get("SQ_INSTS_VALU",0)/tiles, rows.get("SQ_INSTS_LDS",0)/tiles, rows.get("SQ_INSTS_SALU",0)/tiles, rows.get("SQ_LDS_IDX_ACTIVE",0)/tiles, rows.get("SQ_LDS_BANK_CONFLICT",0)/tiles))
PY
done
