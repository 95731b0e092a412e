#!/bin/bash
# Where does fmx::f2::front2_kernel spend its time?  Diagnostic builds with phases compiled out (tools/build_variant.sh
# f2abl<bits> fmx_front2 -DF2_ABL=<bits>; results WRONG by construction): front ms per launch at 4096 channels.
R=$GRAFT_REPO_ROOT
for v in 0 1 2 3 4 7 12 15; do
  lib=$R/sdr-j-fm_amd/lib/ab/libfmx_f2abl$v.so; [ $v = 0 ] && lib=$R/sdr-j-fm_amd/lib/libfmx.so
  [ -f $lib ] || continue
  FMX_LIB=$lib python $R/bench.py --quick --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); print('F2_ABL=$v front ms', j['kernels_ms_per_step']['front_fir'])"
done
