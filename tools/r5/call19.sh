#!/bin/bash
# round 5, GPU call 19: the RCCL path at world size 1 on the final tree (process group, shard ranges, all-gather overlapped)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05m; mkdir -p $O
GNNRAG_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-e2e --spread-steps 0 --fp32-steps 0 > $O/bench_force_dist.log 2> $O/bench_force_dist.err; echo "rc=$?"
tail -1 $O/bench_force_dist.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['n_ranks_seen'], d['rccl_version'], d['config'].get('parallelism'), {k:v for k,v in d.items() if 'gather' in k or 'collective' in k})"
timeout 200 python bench.py --gpus 1 --workload C1 --no-cpu-baseline --no-e2e --spread-steps 0 --fp32-steps 0 --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('C1', round(d['ms_per_step'],4))"
