#!/bin/bash
mkdir -p gpurun_out/r4c6
bash tools/profile_e2e.sh d200 16 2>&1 | tee gpurun_out/r4c6/prof_d200_b16.txt | tail -95
GNNRAG_TUNE_GEMM=1 GNNRAG_TUNE_ONLY=upd python tools/tune_variants.py --run default desync40 desync80 desync160 prio2 prio2_desync80 default desync80 prio2 2>&1 | tee gpurun_out/r4c6/tune_upd.txt
