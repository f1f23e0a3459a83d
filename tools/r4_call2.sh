#!/bin/bash
# round 4, GPU call 2: k_update_b3w (12 / 16 waves per workgroup): parity subset per variant, then timings
mkdir -p gpurun_out/r4c2
for v in updw12 updw16; do
  export GNNRAG_LIB=$PWD/gnn-rag_amd/lib/exp_$v.so
  timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_round3_shapes.py tests/test_gpu_frontier.py "tests/test_gpu_baseline_shapes.py::test_c2_full_batch_against_oracle_slices" 2>&1 | tail -6 | tee gpurun_out/r4c2/pytest_$v.txt
done
unset GNNRAG_LIB
GNNRAG_TUNE_GEMM=1 python tools/tune_variants.py --run default updw12 updw16 default updw12 updw16 2>&1 | tee gpurun_out/r4c2/tune.txt
bash tools/ab_step.sh updw12 updw16 2>&1 | tee gpurun_out/r4c2/ab_step.txt
