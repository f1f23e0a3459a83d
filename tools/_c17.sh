#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/r4c17; mkdir -p $OUT
GNNRAG_TUNE_WORKLOAD=C5 timeout 1200 python tools/tune_variants.py --run default hub_u4 hub_u3 hub_ks16 hub_ks16_u4 hub_ks4_u4 default light_nogather light_nostore light_nomem > $OUT/tune_C5.txt 2>&1
cat $OUT/tune_C5.txt
