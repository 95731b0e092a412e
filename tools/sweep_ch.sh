#!/bin/bash
# per-stage ms per step over channel counts: tools/sweep_ch.sh <lib tag|default> <channels> <channels> ...
R=$GRAFT_REPO_ROOT; v=$1; shift
lib=$R/sdr-j-fm_amd/lib/ab/libfmx_$v.so; [ $v = default ] && lib=$R/sdr-j-fm_amd/lib/libfmx.so
for c in "$@"; do
  FMX_LIB=$lib python $R/bench.py --quick --channels $c 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); k = j['kernels_ms_per_step']
print('%-8s ch %5d  A %.4f  B %.4f  C %.4f  step %.4f  GS/s %.1f' % ('$v', $c, k['front_fir'], k['demod_pilot_pss'], k['audio_fir_resample'], j['ms_per_step'], j['value'] / 1e3))"
done
