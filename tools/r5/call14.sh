#!/bin/bash
# round 5, GPU call 14: host threads limited to the CPU quota - forward sections, main.py parity tests, default bench (e2e block)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max)  nproc: $(nproc)" > $O/forward_host_time.txt
for HT in "" 0; do
  for cfgs in "d200 64 4 8" "d200 16 8 30" "d50 1 8 30"; do
    set -- $cfgs
    echo "== variant $1, test_batch_size $2, GNNRAG_HOST_THREADS='$HT' ('' = limit_host_threads(), 0 = torch's default: one OpenMP thread per visible hardware thread)" >> $O/forward_host_time.txt
    grep nr_throttled /sys/fs/cgroup/cpu.stat | tr '\n' ' ' >> $O/forward_host_time.txt
    timeout 300 bash tools/profile_forward.sh $1 $2 sections GNNRAG_PROFILE_WARM=$3 GNNRAG_PROFILE_CALLS=$4 GNNRAG_HOST_THREADS=$HT 2>&1 | grep -v "INFO\|it/s\|it\]\|GNNRAG_E2E\|collections inside" | tail -14 | cut -c1-400 >> $O/forward_host_time.txt
    grep nr_throttled /sys/fs/cgroup/cpu.stat | tr '\n' ' ' >> $O/forward_host_time.txt; echo >> $O/forward_host_time.txt
  done
done
cat $O/forward_host_time.txt | grep "==\|steady\|nr_thr"
timeout 900 python -m pytest tests/test_gpu_main_py.py -x -q > $O/pytest_main_py.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_main_py.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -1 $O/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['roofline']['frac'], d['step_ms_spread'], d['cpu_baseline']['cores'], d['cpu_baseline']['seconds_per_pass'])
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        for leg,x in v.items():
            if isinstance(x,dict): print(k,leg,round(x['questions_per_s'],1),x['threads'],x['stages_ms_per_batch'])"
