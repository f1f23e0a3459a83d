cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for SK in 0 1 2 4 6 7 15; do
export GNNRAG_SL_SKIP=$SK
timeout 300 rocprofv3 --kernel-trace -d /tmp/tr_$SK -o bench -- python $R/bench.py --workload C3 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --spread-steps 0 --fp32-steps 0 --clock-ramp-ms 100 > /tmp/tr_$SK.log 2>&1
python - /tmp/tr_$SK $SK <<'PY' 2>&1 | head -5
import sqlite3,sys,glob,re
db=glob.glob(sys.argv[1]+'/**/bench_results.db',recursive=True)[0]
c=sqlite3.connect(db)
rows=c.execute("""select s.kernel_name,d.start,d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start""").fetchall()
idx=[i for i,r in enumerate(rows) if 'k_tables_small' in r[0]]
i0=idx[len(idx)//2]
print("skip", sys.argv[2], [round((r[2]-r[1])/1e3,1) for r in rows[i0:i0+4]])
PY
rm -rf /tmp/tr_$SK
done
