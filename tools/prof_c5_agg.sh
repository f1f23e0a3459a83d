# rocprofv3 kernel stats of the fused aggregation alone (dense and seed priors alternate):  bash tools/prof_c5_agg.sh [workload]
W=${1:-C5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_agg_$W -o agg -- python $R/tools/prof_ops.py --workload $W --ops aggf --reps 6 > $R/gpurun_out/prof_agg_$W.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_agg_$W -name "agg_results.db" | head -1) > gpurun_out/stats_agg_$W.txt 2>&1
find gpurun_out/prof_agg_$W -name "*.db" -delete
head -16 gpurun_out/stats_agg_$W.txt | cut -c1-64,92-150
