#!/bin/bash
# round 5, GPU call 4: k_update_b3 with scalar-base + 32-bit-offset addressing (lib/libgnnrag_hip.so) against the previous
# build (lib/exp_default.so); parity of the update tests first
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round3_shapes.py -x -q -k "self_block_update" > $O/pytest_upd.log 2>&1; echo "pytest_upd rc=$?"
tail -2 $O/pytest_upd.log
GNNRAG_TUNE_ONLY=upd timeout 900 python tools/tune_variants.py --run x32_off mainlib x32_off mainlib x32_off mainlib > $O/tune_upd.log 2>&1
cat $O/tune_upd.log
for i in 1 2; do
  timeout 400 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_new_$i.json 2> $O/bench_new_$i.err
  GNNRAG_LIB=$PWD/gnn-rag_amd/lib/exp_default.so GNNRAG_UPDATE_X32=0 timeout 400 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_old_$i.json 2> $O/bench_old_$i.err
done
for f in $O/bench_new_1.json $O/bench_old_1.json $O/bench_new_2.json $O/bench_old_2.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],4), d.get('ms_per_step_fp32'), {k:round(v,4) for k,v in d.get('kernel_ms',{}).items() if 'fused' in k or 'tables' in k})"; done
