#!/bin/bash
# round 4, GPU call 4: k_update_wr with static stages (no vmcnt(0) per tile); main.py tests on the learnable dataset
mkdir -p gpurun_out/r4c4
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_round3_shapes.py tests/test_gpu_frontier.py "tests/test_gpu_baseline_shapes.py::test_c2_full_batch_against_oracle_slices" 2>&1 | tail -6 | tee gpurun_out/r4c4/pytest_wr.txt
GNNRAG_TUNE_GEMM=1 python tools/tune_variants.py --run default wr_off default wr_off 2>&1 | tee gpurun_out/r4c4/tune.txt
export BENCH_SKIP_STRUCTURE_TIMING=1
for r in 1 2; do
  for v in on off; do
    if [ $v = off ]; then export GNNRAG_UPDATE_WR=0; else unset GNNRAG_UPDATE_WR; fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('wr_$v', round(d['ms_per_step'],4), d.get('ms_per_step_fp32'), d['roofline']['measured_copy_ceiling_GBps'])
" | tee -a gpurun_out/r4c4/ab_step.txt
  done
done
unset GNNRAG_UPDATE_WR
timeout 1200 python -m pytest -x -q -m gpu tests/test_gpu_main_py.py -k "d50 or cwq" -s 2>&1 | grep -v "^$" | tail -12 | tee gpurun_out/r4c4/pytest_main_py.txt
