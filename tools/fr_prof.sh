cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in C2 C1 C3; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fr_$W -o fr -- python $R/tools/prof_ops.py --workload $W --reps 20 --ops fr > $R/gpurun_out/fr_$W.log 2>&1
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/prof_fr_$W -name "fr_results.db" | head -1) > $R/gpurun_out/fr_stats_$W.txt 2>&1
grep -n "frontier\|calls" $R/gpurun_out/fr_stats_$W.txt
done
find $R/gpurun_out -name "*.db" -delete
