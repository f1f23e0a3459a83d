#!/bin/bash
# round 5, GPU call 7: the final tree - whole GPU suite, smoke, the default bench line as the driver runs it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -1 $O/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['value'], d['roofline']['frac'], d['cpu_baseline'])"
