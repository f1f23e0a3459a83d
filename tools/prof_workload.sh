# rocprofv3 kernel stats of one workload's bench run:  bash tools/prof_workload.sh C1 [extra bench flags]
W=${1:-C1}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$W -o bench -- python $R/bench.py --workload $W --no-cpu-baseline --steps 30 --spread-steps 0 --fp32-steps 0 "$@" > $R/gpurun_out/prof_$W.log 2>&1
cd $R
python tools/rocpd_stats.py $(find gpurun_out/prof_$W -name "bench_results.db" | head -1) > gpurun_out/stats_$W.txt 2>&1
find gpurun_out/prof_$W -name "*.db" -delete
head -24 gpurun_out/stats_$W.txt | cut -c1-60,92-150
tail -1 gpurun_out/prof_$W.log | cut -c1-200
