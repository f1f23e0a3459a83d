#!/bin/bash
# round 5, GPU call 1: parity of the 32x32x16 update kernel, its A/B against k_update_b3, the LDS-walk variants
# (named stages, 640 / 768 threads), a bench line with and without it, then the whole GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3_shapes.py -x -q -k "self_block_update" > $O/pytest_upd.log 2>&1; echo "pytest_upd rc=$?"
tail -3 $O/pytest_upd.log
GNNRAG_TUNE_ONLY=upd timeout 600 python tools/tune_variants.py --run default x32_off default x32_off > $O/tune_upd.log 2>&1
cat $O/tune_upd.log
timeout 1200 python tools/tune_variants.py --run default sl_named sl_named_t768 sl_named_t640 sl_named_t640_g2 sl_t768 sl_t640 default > $O/tune_walk.log 2>&1
cat $O/tune_walk.log
for i in 1 2; do
  timeout 400 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_x32_$i.json 2> $O/bench_x32_$i.err
  GNNRAG_UPDATE_X32=0 timeout 400 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_b3_$i.json 2> $O/bench_b3_$i.err
done
for f in $O/bench_x32_1.json $O/bench_b3_1.json $O/bench_x32_2.json $O/bench_b3_2.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],4), d.get('ms_per_step_fp32'), {k:round(v,4) for k,v in d.get('kernel_ms',{}).items()})"; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?"
tail -5 $O/pytest_all.log
