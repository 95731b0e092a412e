#!/bin/bash
# Diagnostic variant of libfmx.so: one source rebuilt with extra -D flags, the other objects as built.
#   tools/build_variant.sh <tag> <source stem, e.g. fmx_front2> -DF2_ABL=1 ...   ->  sdr-j-fm_amd/lib/ab/libfmx_<tag>.so   (run with FMX_LIB=...)
R=$(cd $(dirname $0)/.. && pwd); L=$R/sdr-j-fm_amd/lib; TAG=$1; SRC=$2; shift; shift
mkdir -p $L/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $( [ $SRC = fmx_front -o $SRC = fmx_front4 -o $SRC = fmx_front4lo -o $SRC = fmx_audio ] || echo -ffp-contract=off ) "$@" -c $R/sdr-j-fm_amd/csrc/$SRC.hip -o $L/ab/${SRC}_$TAG.o || exit 1
OBJS=""
for o in fmx_front fmx_front4 fmx_front4lo fmx_demod fmx_stageb fmx_audio fmx_rds fmx_ola fmx_promote fmx_api; do
  if [ $o = $SRC ]; then OBJS="$OBJS $L/ab/${SRC}_$TAG.o"; else OBJS="$OBJS $L/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/ab/libfmx_$TAG.so $OBJS
echo $L/ab/libfmx_$TAG.so
