#!/bin/bash
# kernel timeline of one bench step (on the GPU box): tools/prof_timeline.sh <name> [bench args...]
R=$GRAFT_REPO_ROOT; N=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$N -o $N -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/bench_$N.log 2>&1
cd $R
python tools/timeline.py gpurun_out/prof_$N/${N}_results.db 2 > gpurun_out/timeline_$N.txt
rm -rf gpurun_out/prof_$N
