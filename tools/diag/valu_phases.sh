# VALU instructions per launch of stage B's first kernel with one phase compiled out (-DSB_ABLATE=bit): the difference to the full
# kernel is that phase's share.  Build: for v in 1 2 4 8; do tools/build_variant.sh abl$v fmx_stageb -DSB_ABLATE=$v -ffp-contract=off; done
cd $GRAFT_REPO_ROOT
for v in default abl1 abl2 abl4 abl8; do
  lib=$GRAFT_REPO_ROOT/sdr-j-fm_amd/lib/ab/libfmx_$v.so; [ $v = default ] && lib=$GRAFT_REPO_ROOT/sdr-j-fm_amd/lib/libfmx.so
  echo -n "$v: "; FMX_LIB=$lib bash tools/pmc_kernel.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "stageb_kernel<1>" --quick --steps 4 --warmup 44 2>&1 | tail -1
done
