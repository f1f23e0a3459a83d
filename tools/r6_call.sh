cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=(
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
 "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"
)
for mode in regular lite; do
  i=0
  for C in "${SETS[@]}"; do
    i=$((i+1))
    if [ $mode = lite ]; then export GNNRAG_TABLES_LITE_MAIN=1; else unset GNNRAG_TABLES_LITE_MAIN; fi
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$mode -o set$i -- python $R/tools/prof_ops.py --workload C2 --reps 3 --ops layer > $O/pmc_${mode}_set$i.log 2>&1 || echo "pass $mode $i failed"
  done
  cd $R
  python tools/rocpd_pmc.py $(find $O/pmc_$mode -name '*_results.db' | sort) > $O/pmc_tables_$mode.txt 2>&1
  cd /tmp
done
unset GNNRAG_TABLES_LITE_MAIN
find $O -name '*.db' -delete
cd $R
grep -A40 "tables_vq" $O/pmc_tables_regular.txt | head -45
grep -A40 "tables_vq_lite" $O/pmc_tables_lite.txt | head -45
