#!/bin/bash
# round 5, GPU call 2: form 2 of the 32x32x16 update kernel (one wave per SIMD): parity, then A/B against k_update_b3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
GNNRAG_UPDATE_X32=2 timeout 900 python -m pytest tests/test_gpu_round3_shapes.py -x -q -k "self_block_update" > $O/pytest_upd_f2.log 2>&1; echo "pytest_upd_f2 rc=$?"
tail -3 $O/pytest_upd_f2.log
GNNRAG_TUNE_ONLY=upd timeout 900 python tools/tune_variants.py --run x32_off x32_f2 x32_f1 x1_valu2 x1_valu4 x1_valu6 x32_off x32_f2 > $O/tune_upd.log 2>&1
cat $O/tune_upd.log
for i in 1 2; do
  GNNRAG_UPDATE_X32=2 timeout 400 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_f2_$i.json 2> $O/bench_f2_$i.err
  GNNRAG_UPDATE_X32=0 timeout 400 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_b3_$i.json 2> $O/bench_b3_$i.err
done
for f in $O/bench_f2_1.json $O/bench_b3_1.json $O/bench_f2_2.json $O/bench_b3_2.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],4), d.get('ms_per_step_fp32'), {k:round(v,4) for k,v in d.get('kernel_ms',{}).items() if 'fused' in k or 'tables' in k})"; done
GNNRAG_UPDATE_X32=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py tests/test_device_fact_cache.py tests/test_eval_tail.py -m gpu -x -q > $O/pytest_some.log 2>&1; echo "pytest_some rc=$?"
tail -4 $O/pytest_some.log
