#!/bin/bash
mkdir -p gpurun_out/r4c11
GNNRAG_LIB=$PWD/gnn-rag_amd/lib/exp_vq_order1.so timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_parity.py -k "relation_tables or planes or stack_matches or c2" 2>&1 | tail -4 | tee gpurun_out/r4c11/pytest_vq_order1.txt
GNNRAG_TUNE_GEMM=1 python tools/tune_variants.py --run default vq_order1 default vq_order1 2>&1 | tee gpurun_out/r4c11/tune.txt
bash tools/ab_step.sh vq_order1 2>&1 | tee gpurun_out/r4c11/ab_step.txt
