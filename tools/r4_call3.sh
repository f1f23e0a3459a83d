#!/bin/bash
# round 4, GPU call 3: k_update_wr (weights in registers) parity + timing; module-path graph replay; main.py on the new dataset
mkdir -p gpurun_out/r4c3
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_round3_shapes.py tests/test_gpu_frontier.py "tests/test_gpu_baseline_shapes.py::test_c2_full_batch_against_oracle_slices" tests/test_eval_tail.py 2>&1 | tail -8 | tee gpurun_out/r4c3/pytest_wr.txt
GNNRAG_TUNE_GEMM=1 python tools/tune_variants.py --run default wr_off default wr_off 2>&1 | tee gpurun_out/r4c3/tune.txt
export BENCH_SKIP_STRUCTURE_TIMING=1
for r in 1 2; do
  for v in on off; do
    if [ $v = off ]; then export GNNRAG_UPDATE_WR=0; else unset GNNRAG_UPDATE_WR; fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('wr_$v', round(d['ms_per_step'],4), d.get('ms_per_step_fp32'), d['roofline']['measured_copy_ceiling_GBps'])
" | tee -a gpurun_out/r4c3/ab_step.txt
  done
done
unset GNNRAG_UPDATE_WR
python tools/time_module_graph.py C1 2>&1 | tail -2 | tee gpurun_out/r4c3/module_graph.txt
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_main_py.py -k "d50 or cwq" -s 2>&1 | tail -12 | tee gpurun_out/r4c3/pytest_main_py.txt
