#!/bin/bash
# PMC picture of the self-block update alone, per kernel form (GNNRAG_UPDATE_X32 = 0: k_update_b3, 1 / 2: the 32x32x16 forms)
# usage (GPU box): bash tools/pmc_update.sh <tag> "<forms>"
TAG=${1:-r5c}
FORMS=${2:-"0 2"}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
SETS=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
 "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TA_TA_BUSY_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
for F in $FORMS; do
  i=0
  for C in "${SETS[@]}"; do
    i=$((i+1))
    GNNRAG_UPDATE_X32=$F timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmcu_f$F -o set$i -- python $R/tools/prof_ops.py --workload C2 --reps 3 --ops updfd > $OUT/pmcu_f${F}_set$i.log 2>&1 || echo "form $F pass $i failed"
  done
  ( cd $R && python tools/rocpd_pmc.py $(find $OUT/pmcu_f$F -name '*_results.db' | sort) > $OUT/pmc_update_f$F.txt 2>&1 )
  grep -A40 "k_update" $OUT/pmc_update_f$F.txt | head -60
done
find $OUT -name '*.db' -delete
