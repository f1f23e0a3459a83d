#!/bin/bash
# round 5, GPU call 21 (last): main.py parity tests through the changed run_reference.py, then the default bench as the driver runs it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05n; mkdir -p $O
a=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
timeout 300 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"
b=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); echo "nr_throttled during the default bench: +$((b-a))"
tail -1 $O/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), r['frac'], r['traffic'], d['step_ms_spread']['max'], d['cpu_baseline']['cores'], d['cpu_baseline']['seconds_per_pass'])
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        for leg,x in v.items():
            if isinstance(x,dict): print(k,leg,round(x['questions_per_s'],1),{a:round(b,2) for a,b in x['stages_ms_per_batch'].items()})"
timeout 200 python -m pytest tests/test_gpu_main_py.py -x -q > $O/pytest_main_py.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_main_py.log
