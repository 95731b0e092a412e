#!/bin/bash
# short bench line: tools/bq.sh <bench args...>   (on the GPU box)
timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['config']['channels_per_gpu'], j['kernels_ms_per_step'], 'GS/s %.1f' % (j['value'] / 1e3), 'ms/step', j['ms_per_step'], 'frac', j['roofline']['frac'])"
