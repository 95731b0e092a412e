#!/bin/bash
# the same build in N fresh processes: stage A's time per launch next to where the caller-side buffers landed (on the GPU box)
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-8}); do
  FMX_BENCH_PTRS=1 timeout 300 python bench.py --no-cpu-baseline --quick 2> /tmp/err.txt | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('front %.4f stageb %.4f step %.4f' % (j['roofline']['avg_launch_ms'], j['kernels_ms_per_step']['demod_pilot_pss'], j['ms_per_step']), end=' ')"
  grep ptrs /tmp/err.txt
done
