#!/bin/bash
# round 5, GPU call 12: gc.freeze() after setup - section timers with / without, the main.py parity tests, the default bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
for FZ in 1 0; do
  timeout 300 bash tools/profile_forward.sh d200 64 sections GNNRAG_PROFILE_WARM=4 GNNRAG_PROFILE_CALLS=8 GNNRAG_GC_FREEZE=$FZ 2>&1 | grep -v "INFO\|it/s\|it\]" | tail -11 | cut -c1-600
  timeout 300 bash tools/profile_forward.sh d200 16 sections GNNRAG_GC_FREEZE=$FZ 2>&1 | grep -v "INFO\|it/s\|it\]" | tail -11 | cut -c1-600
  timeout 300 bash tools/profile_forward.sh d50 1 sections GNNRAG_GC_FREEZE=$FZ 2>&1 | grep -v "INFO\|it/s\|it\]" | tail -11 | cut -c1-600
done > $O/forward_sections.txt 2>&1
cat $O/forward_sections.txt
timeout 900 python -m pytest tests/test_gpu_main_py.py -x -q > $O/pytest_main_py.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_main_py.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -1 $O/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['roofline']['frac'])
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        for leg,x in v.items():
            if isinstance(x,dict): print(k,leg,round(x['questions_per_s'],1),x['stages_ms_per_batch'])"
