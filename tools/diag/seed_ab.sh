cd $GRAFT_REPO_ROOT
for v in default seed1 seed0; do
  lib=sdr-j-fm_amd/lib/ab/libfmx_$v.so; [ $v = default ] && lib=sdr-j-fm_amd/lib/libfmx.so
  echo "== $v"; FMX_LIB=$GRAFT_REPO_ROOT/$lib python tools/stageb_rounds.py 1024 48 2>&1 | grep "PLL rounds"
done
bash tools/cmp_variants.sh seed1 seed0
for v in seed1 seed0; do FMX_LIB=$GRAFT_REPO_ROOT/sdr-j-fm_amd/lib/ab/libfmx_$v.so python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -4; done
