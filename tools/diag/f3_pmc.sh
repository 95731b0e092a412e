#!/bin/bash
# PMC counters of the two stage-A kernels side by side (on the GPU box): tools/diag/f3_pmc.sh [bench args]
cd $GRAFT_REPO_ROOT
for fk in 2 1; do
  kn=front3_kernel; [ $fk = 1 ] && kn=front_kernel
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
             "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT" \
             "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
    FMX_FRONT_KERNEL=$fk bash tools/pmc_kernel.sh "$set" $kn --quick --steps 4 --warmup 44 "$@" 2>&1 | tail -1
  done
done
