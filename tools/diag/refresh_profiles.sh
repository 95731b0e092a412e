#!/bin/bash
# the round's judged artefacts in one GPU call: default bench line, kernel stats of config4 and config5
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_err.log
tail -c 6000 gpurun_out/r03_bench_line.json
bash tools/prof.sh r03_config4 --no-cpu-baseline --no-extra-workloads > /dev/null 2>&1
bash tools/prof.sh r03_config5 --workload config5 --no-cpu-baseline --no-extra-workloads > /dev/null 2>&1
head -14 gpurun_out/r03_config4_kernel_stats.txt | cut -c1-200
head -24 gpurun_out/r03_config5_kernel_stats.txt | cut -c1-200
