#!/bin/bash
# Diagnostic variant of libfmx.so: fmx_front.hip rebuilt with extra -D flags, the other objects as built.
#   tools/build_variant.sh <tag> -DFMX_FRONT_VALU_FIR=1 ...   ->  sdr-j-fm_amd/lib/ab/libfmx_<tag>.so   (run with FMX_LIB=...)
R=$(cd $(dirname $0)/.. && pwd); L=$R/sdr-j-fm_amd/lib; TAG=$1; shift
mkdir -p $L/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $R/sdr-j-fm_amd/csrc/fmx_front.hip -o $L/ab/front_$TAG.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/ab/libfmx_$TAG.so $L/ab/front_$TAG.o $L/fmx_demod.o $L/fmx_stageb.o $L/fmx_audio.o $L/fmx_rds.o $L/fmx_api.o
echo $L/ab/libfmx_$TAG.so
