#!/bin/bash
# front4_kernel with phases compiled out (F4_ABL bits: 1 scatter, 2 DC pass, 4 matrix FIR, 8 tile loads behind the first, 16 output stores): build here with
#   for v in 7 23 8 24 16 4 1 2; do bash tools/build_variant.sh f4abl$v fmx_front4 -DF4_ABL=$v; done
# then on the GPU box: tools/diag/f4_ablate.sh     (only the front_fir column means anything: the later stages get garbage)
cd $GRAFT_REPO_ROOT
echo -n "default: "; bash tools/bq.sh --quick
for f in sdr-j-fm_amd/lib/ab/libfmx_f4abl*.so; do
  echo -n "$(basename $f): "; FMX_LIB=$GRAFT_REPO_ROOT/$f bash tools/bq.sh --quick
done
echo -n "default: "; bash tools/bq.sh --quick
