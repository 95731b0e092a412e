#!/bin/bash
# kernel times of the demodulator pre-pass workloads (PLL decoder, AM decoder, noise squelch, level squelch on every channel): on the GPU box
R=$GRAFT_REPO_ROOT
for w in "--decoder 2" "--decoder 1" "--squelch 1" "--squelch 2"; do
  tag=pp_$(echo $w | tr -d ' -')
  bash $R/tools/prof.sh $tag --no-cpu-baseline $w > /dev/null 2>&1
  echo "== bench.py --quick $w"; head -12 $R/gpurun_out/${tag}_kernel_stats.txt | cut -c1-170
done
