#!/bin/bash
# the round's judged artefacts in one GPU call: default bench line, kernel stats of config4 / mixed population / config5, stage A's HBM
# traffic (PMC), the counters of stage A and of stage B's two kernels.  Everything lands in gpurun_out/r06_*; copy into profiles/.
cd $GRAFT_REPO_ROOT
# (the headline's kernel trace first, on a box that has done nothing yet: the stage-A kernel's duration follows the chip's power state, and a trace taken behind
# the five-minute default line meets a hotter chip than the line's own timed region did)
bash tools/prof.sh r06_config4 > /dev/null 2>&1
sleep 20
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_err.log
tail -c 3000 gpurun_out/r06_bench_line.json
bash tools/prof.sh r06_mixed --population mixed > /dev/null 2>&1
bash tools/prof.sh r06_config5 --workload config5 > /dev/null 2>&1
bash tools/prof.sh r06_config3 --workload config3 > /dev/null 2>&1
bash tools/prof.sh r06_config1 --workload config1 > /dev/null 2>&1
bash tools/prof.sh r06_pll_decoder --decoder 2 > /dev/null 2>&1
bash tools/prof.sh r06_noise_squelch --squelch 1 > /dev/null 2>&1
bash tools/prof.sh r06_level_squelch --squelch 2 > /dev/null 2>&1
bash tools/pmc_traffic.sh r06 --quick --steps 4 --warmup 44 2>&1 | tail -1
{
  echo "# rocprofv3 --pmc passes of python bench.py --quick (4096 channels, established population), averages per launch and per collection unit"
  for set in "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAVES SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    for kn in "front4_kernel" "stageb_kernel<0>" "audio_fft"; do
      bash tools/pmc_kernel.sh "$set" "$kn" --quick --steps 4 --warmup 44 2>&1 | tail -1
    done
  done
} > gpurun_out/r06_counters.txt
cat gpurun_out/r06_counters.txt
head -12 gpurun_out/r06_config4_kernel_stats.txt | cut -c1-160
head -14 gpurun_out/r06_mixed_kernel_stats.txt | cut -c1-160
