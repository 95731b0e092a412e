#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench run on the GPU box, summary into gpurun_out/<tag>_kernel_stats.txt
#   tools/prof.sh <tag> [bench args...]
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py --quick "$@" > /tmp/prof_$TAG.log 2>&1
DB=$(find /tmp/prof_$TAG -name "${TAG}_results.db" | head -1)
python $R/tools/rocprof_summary.py $DB /tmp/prof_$TAG.log > $R/gpurun_out/${TAG}_kernel_stats.txt
head -40 $R/gpurun_out/${TAG}_kernel_stats.txt
