#!/bin/bash
# per-stage ms per step of the default build and of the diagnostic variants named: tools/cmp_variants.sh <tag> <tag> ... [-- bench args]
R=$GRAFT_REPO_ROOT; TAGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do TAGS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for v in default "${TAGS[@]}"; do
  lib=$R/sdr-j-fm_amd/lib/ab/libfmx_$v.so; [ $v = default ] && lib=$R/sdr-j-fm_amd/lib/libfmx.so
  [ -f $lib ] || { echo "no $lib"; continue; }
  for rep in 1 2; do
  FMX_LIB=$lib python $R/bench.py --quick --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); k = j['kernels_ms_per_step']
print('%-12s A %.4f  B %.4f  C %.4f  step %.4f  GS/s %.1f' % ('$v', k['front_fir'], k['demod_pilot_pss'], k['audio_fir_resample'], j['ms_per_step'], j['value'] / 1e3))"
  done
done
