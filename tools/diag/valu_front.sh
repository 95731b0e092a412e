# VALU / LDS instructions per launch of front_kernel with one phase compiled out (-DFMX_ABL=bit: 1 scatter, 2 DC pass, 4 FIR)
cd $GRAFT_REPO_ROOT
for v in default fabl1 fabl2 fabl4; do
  lib=$GRAFT_REPO_ROOT/sdr-j-fm_amd/lib/ab/libfmx_$v.so; [ $v = default ] && lib=$GRAFT_REPO_ROOT/sdr-j-fm_amd/lib/libfmx.so
  echo -n "$v: "; FMX_LIB=$lib bash tools/pmc_kernel.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "front_kernel" --quick --steps 4 --warmup 44 2>&1 | tail -1
done
