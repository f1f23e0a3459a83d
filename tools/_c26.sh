#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r04f2; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -x > $OUT/gpu_tests.log 2>&1; tail -4 $OUT/gpu_tests.log | cut -c1-300
cp profiles/pmc_traffic.json /tmp/pmc_c2.json
cp profiles/r04f_pmc_traffic_C5.json profiles/pmc_traffic.json
python bench.py --workload C5 --no-cpu-baseline --steps 30 > $OUT/bench_C5.log 2>&1; tail -1 $OUT/bench_C5.log | cut -c1-200
cp /tmp/pmc_c2.json profiles/pmc_traffic.json
