cd $GRAFT_REPO_ROOT
for s in 0 1 0 1; do echo "SPLIT=$s"; FMX_STAGEB_SPLIT=$s bash tools/cmp_variants.sh 2>&1 | head -2; done
FMX_STAGEB_SPLIT=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
