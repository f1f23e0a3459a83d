cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06a/gpu_tests.log 2>&1
tail -3 gpurun_out/r06a/gpu_tests.log
( time python bench.py ) > gpurun_out/r06a/bench_default.log 2> gpurun_out/r06a/bench_default.err
echo rc=$?
tail -5 gpurun_out/r06a/bench_default.err
tail -1 gpurun_out/r06a/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d.get('parity_in_run'))[:1500])
print(json.dumps({k:{kk:vv for kk,vv in v.items() if kk in ('ms_per_step','value','error')} | {'frac':(v.get('roofline') or {}).get('frac')} for k,v in d.get('other_workloads',{}).items()}))
print(json.dumps(d.get('train_step'))[:3000])
"
