#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel stats + the two PMC passes the
# roofline's `traffic` figure comes from.  Usage: tools/refresh_profiles.sh <tag>   (e.g. r01i)
# Outputs under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_default.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc -o $C -- python $R/tools/prof_ops.py --workload C2 --reps 4 --ops agg,aggf > $OUT/pmc_$C.log 2>&1
done
cd $R
python tools/rocpd_stats.py $(find $OUT/trace -name 'bench_results.db' | head -1) > $OUT/kernel_stats_bench.txt 2>&1
python tools/make_pmc_traffic.py $(find $OUT/pmc -name 'FETCH_SIZE_results.db' | head -1) $(find $OUT/pmc -name 'WRITE_SIZE_results.db' | head -1) $OUT/pmc_traffic.json > $OUT/make_pmc.log 2>&1
find $OUT -name '*.db' -size +20M -delete
tail -1 $OUT/bench_default.log | cut -c1-400
head -12 $OUT/kernel_stats_bench.txt
