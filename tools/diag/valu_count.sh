# VALU instructions per launch of the two stage-B kernels (deterministic: a better yardstick than time for small changes)
cd $GRAFT_REPO_ROOT
for kn in "stageb_kernel<1>" "stageb_kernel<2>"; do
  bash tools/pmc_kernel.sh "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "$kn" --quick --steps 4 --warmup 44 2>&1 | tail -1
done
