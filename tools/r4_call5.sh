#!/bin/bash
# round 4, GPU call 5: prefetcher / graph / wr-opt-in tests, main.py on all variants, update-kernel phase experiments,
# one full default bench line (with the e2e block)
mkdir -p gpurun_out/r4c5
timeout 1500 python -m pytest -x -q -m gpu tests/test_device_fact_cache.py tests/test_gpu_round3_shapes.py tests/test_gpu_main_py.py -k "not d200" -s 2>&1 | grep -v "^$" | tail -25 | tee gpurun_out/r4c5/pytest.txt
GNNRAG_TUNE_GEMM=1 GNNRAG_TUNE_ONLY=upd python tools/tune_variants.py --run default desync40 desync80 desync160 prio2 prio2_desync80 default desync80 prio2 2>&1 | tee gpurun_out/r4c5/tune_upd.txt
python bench.py > gpurun_out/r4c5/bench_default.log 2>&1; tail -1 gpurun_out/r4c5/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'frac_incl_tables', d['roofline'].get('frac_incl_tables'), 'copy', d['roofline']['measured_copy_ceiling_GBps'])
print('overlap', json.dumps(d.get('structure_build_overlapped')), 'csr_build_ms', d['csr_build_ms'], d['csr_build_from_device_cache_ms'])
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:600])
print('e2e', json.dumps(d.get('e2e'))[:3000])
print('upd', json.dumps(d['roofline_dense']['update_score_fused'])[:800])
"
