#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r4c18; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hub_rows.py tests/test_gpu_baseline_shapes.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
GNNRAG_TUNE_WORKLOAD=C5 timeout 1200 python tools/tune_variants.py --run default quad_off quad_s1 quad_s3 quad_s4 quad_npw1 quad_npw4 default > $OUT/tune_C5.txt 2>&1
cat $OUT/tune_C5.txt
