# PMC picture of the fused LDS walk (streaming form vs the set walk): where do the waves wait?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for MODE in stream sets; do
  if [ $MODE = stream ]; then export GNNRAG_WALK_STREAM=1; else unset GNNRAG_WALK_STREAM; fi
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    T=$(echo $C | cut -d" " -f1)
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmcw_$MODE -o $T -- python $R/tools/prof_ops.py --workload C2 --reps 3 --ops aggf > $R/gpurun_out/pmcw_${MODE}_$T.log 2>&1
  done
done
cd $R
python - <<'PY'
import sqlite3, glob, os
for mode in ("stream", "sets"):
    for db in sorted(glob.glob("gpurun_out/pmcw_%s/**/*_results.db" % mode, recursive=True)):
        con = sqlite3.connect(db)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")]
        info = [t for t in tabs if t.startswith("rocpd_info_pmc")]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
        ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
        if not (pmc and info and kd and ks):
            print(db, "tables?", tabs[:8]); continue
        q = """select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from %s e join %s i on e.pmc_id = i.id
               join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id
               where s.kernel_name like '%%k_walk_s%%' group by s.kernel_name, i.name""" % (pmc[0], info[0], kd[0], ks[0])
        try:
            for name, c, v, n in con.execute(q):
                print(mode, name[:40], c, "%.4g per launch" % (v / max(n, 1)))
        except Exception as ex:
            print(db, "query failed", ex)
PY
find gpurun_out -name "*.db" -delete
