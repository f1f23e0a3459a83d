#!/bin/bash
# PMC picture of one dense-prior layer at a workload (k_tables_vq, k_fact_prior_merged, k_walk_slice, k_update_b3 /
# their round-4 successors): instruction mix, wave states, MFMA-busy and the clock the chip held.  One pass per counter
# set (gfx950: 8 SQ slots per pass; never combined with a trace domain other than --kernel-trace).
# Usage (GPU box): bash tools/pmc_dense.sh <tag> [workload]
set -u
TAG=${1:-rXX}
W=${2:-C2}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
 "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for C in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmcd_$W -o set$i -- python $R/tools/prof_ops.py --workload $W --reps 3 --ops layer > $OUT/pmcd_${W}_set$i.log 2>&1 || echo "pass $i failed (see $OUT/pmcd_${W}_set$i.log)"
done
cd $R
python tools/rocpd_pmc.py $(find $OUT/pmcd_$W -name '*_results.db' | sort) > $OUT/pmc_dense_$W.txt 2>&1
find $OUT -name '*.db' -delete
grep -v "^ *$" $OUT/pmc_dense_$W.txt | head -150
