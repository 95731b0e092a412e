# rocprofv3 median times (ns) of stage B's kernels for builds side by side on one box: tools/diag/k12_times.sh default <variant tag> ...
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
  lib=$R/sdr-j-fm_amd/lib/ab/libfmx_$v.so; [ $v = default ] && lib=$R/sdr-j-fm_amd/lib/libfmx.so
  rm -rf /tmp/pk; FMX_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/pk -o p -- python $R/bench.py --quick > /dev/null 2>&1
  echo -n "$v: "; python $R/tools/rocprof_summary.py $(find /tmp/pk -name "p_results.db" | head -1) 2>/dev/null | grep -E "^void fmx::stageb_kernel.* n=" | sed 's/void fmx:://' | awk '{printf "%s %s %s  ", $1, $4, $6}'; echo
done; done
