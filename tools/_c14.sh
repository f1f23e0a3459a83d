#!/bin/bash
set -u
R=$PWD; OUT=$R/gpurun_out/r4c16; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hub_rows.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python $R/tools/prof_ops.py --workload C5 --reps 10 --ops aggfd > $OUT/prof.log 2>&1
cd $R; tail -2 $OUT/prof.log
python tools/rocpd_stats.py $(find $OUT/tr -name 't_results.db' | head -1) > $OUT/stats.txt 2>&1; head -14 $OUT/stats.txt | cut -c1-150
find $OUT -name '*.db' -delete
python bench.py --workload C5 --no-cpu-baseline --no-e2e --steps 20 > $OUT/bench_C5.log 2>&1; tail -1 $OUT/bench_C5.log | cut -c1-400
