#!/bin/bash
# round 5, GPU call 8: the (segment, relation) key sort of the hub rows - structure tests, C5 / C2fb build times with
# both forms, kernel trace of a C5 bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_device_fact_cache.py tests/test_gpu_baseline_shapes.py -x -q > $O/pytest_csr.log 2>&1; echo "pytest rc=$?"
tail -4 $O/pytest_csr.log
for W in C5 C2fb; do
  for FORM in keys segmented; do
    GNNRAG_HUB_SORT=$FORM timeout 600 python bench.py --workload $W --no-cpu-baseline --no-e2e --steps 20 --spread-steps 0 > $O/bench_${W}_$FORM.json 2> $O/bench_${W}_$FORM.err
    echo -n "$W $FORM: "; tail -1 $O/bench_${W}_$FORM.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['csr_build_ms'],3), d['csr_build_from_device_cache_ms'], d.get('value_incl_upload_and_build'))"
  done
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/trace_C5 -o bench -- python $R/bench.py --workload C5 --no-cpu-baseline --no-e2e --steps 10 --spread-steps 0 --clock-ramp-ms 0 > $R/$O/bench_C5_under_rocprof.log 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/trace_C5 -name 'bench_results.db' | head -1) > $O/kernel_stats_bench_C5.txt 2>&1
head -14 $O/kernel_stats_bench_C5.txt | cut -c1-170
rm -rf $O/trace_C5
