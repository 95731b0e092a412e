#!/bin/bash
# kernel-trace stats of one bench run (on the GPU box): tools/prof_kernels.sh <name> [bench args...]
R=$GRAFT_REPO_ROOT; N=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$N -o $N -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/bench_$N.log 2>&1
cd $R
python tools/rocprof_summary.py gpurun_out/prof_$N/${N}_results.db gpurun_out/bench_$N.log > gpurun_out/summary_$N.txt
grep "fmx::" gpurun_out/summary_$N.txt | head -14
rm -rf gpurun_out/prof_$N
