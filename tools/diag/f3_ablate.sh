#!/bin/bash
# front3_kernel with phases compiled out (F3_ABL bits: 1 scatter, 2 DC pass, 4 FIR, 8 tile loads, 16 waits): build here with
#   for v in 1 2 4 8 16 6 7 23; do bash tools/build_variant.sh abl$v fmx_front3 -DF3_ABL=$v; done
# then on the GPU box: tools/diag/f3_ablate.sh
cd $GRAFT_REPO_ROOT
echo -n "default: "; bash tools/bq.sh --quick
for f in sdr-j-fm_amd/lib/ab/libfmx_abl*.so; do
  echo -n "$(basename $f): "; FMX_LIB=$GRAFT_REPO_ROOT/$f bash tools/bq.sh --quick
done
echo -n "front_kernel: "; FMX_FRONT_KERNEL=1 bash tools/bq.sh --quick
