#!/bin/bash
# SQ counters of the attention forward kernel (two passes of 8 SQ counters), summarised per kernel by tools/pmc_kernel.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  case $i in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS";;
    2) C="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE";;
    3) C="SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES GRBM_GUI_ACTIVE";;
  esac
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_attn4_$i -- python3 $R/tools/prof_attn.py > $R/gpurun_out/pmc_attn4_$i.log 2>&1
  python3 $R/tools/pmc_kernel.py $R/gpurun_out/pmc_attn4_$i attn_fwd 70
done
