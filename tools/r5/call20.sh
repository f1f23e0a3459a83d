#!/bin/bash
# round 5, GPU call 20: host stalls inside the timed region - short default-config bench runs from a cold box, the thread pools
# of the process limited before numpy / torch are imported (default) against the previous behaviour (GNNRAG_HOST_THREADS=0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05n; mkdir -p $O
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max)  nproc: $(nproc)" > $O/host_stall.txt
for i in 1 2 3; do
  for HT in 0 ""; do
    a=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
    GNNRAG_HOST_THREADS=$HT timeout 200 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 --fp32-steps 0 > $O/run_${i}_ht${HT:-default}.json 2>/dev/null
    b=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
    echo "run $i GNNRAG_HOST_THREADS='${HT}': ms_per_step $(tail -1 $O/run_${i}_ht${HT:-default}.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), 'walk us', round(d['kernel_ms']['aggregate_fused_dense']*1e3,1))")  nr_throttled +$((b-a))" >> $O/host_stall.txt
  done
done
cat $O/host_stall.txt
