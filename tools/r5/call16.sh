#!/bin/bash
# round 5, GPU call 16: QueryReform as one launch - parity tests, closed loop, main.py parity, forward sections, e2e legs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_closed_loop.py tests/test_gpu_main_py.py -x -q -s > $O/pytest_qr.log 2>&1; echo "pytest rc=$?"; grep "QueryReform at C2\|passed\|failed" $O/pytest_qr.log
for cfgs in "d200 64 4 8" "d200 16 8 30" "d50 1 8 30"; do
  set -- $cfgs
  timeout 300 bash tools/profile_forward.sh $1 $2 sections GNNRAG_PROFILE_WARM=$3 GNNRAG_PROFILE_CALLS=$4 2>&1 | grep "steady\|QueryReform\|ReasonGNN" | cut -c1-300
done
timeout 900 python bench.py --steps 20 --spread-steps 0 --fp32-steps 0 --cpu-sample-b 4 > $O/bench_e2e.json 2> $O/bench_e2e.err; echo "bench rc=$?"
tail -1 $O/bench_e2e.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        for leg,x in v.items():
            if isinstance(x,dict): print(k,leg,round(x['questions_per_s'],1),{a:round(b,2) for a,b in x['stages_ms_per_batch'].items()}, x['test_f1_h1'])"
