#!/bin/bash
# round 5, GPU call 17: the final tree once more after the QueryReform launch - whole GPU suite, smoke, default bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GNNRAG_COMMIT=$(cat .commit_stamp 2>/dev/null || echo unknown)
O=gpurun_out/r05l; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep "passed\|failed" $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"
tail -1 $O/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), r['frac'], r['traffic'], r.get('real_traffic_frac'))
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        for leg,x in v.items():
            if isinstance(x,dict): print(k,leg,round(x['questions_per_s'],1),{a:round(b,2) for a,b in x['stages_ms_per_batch'].items()})"
