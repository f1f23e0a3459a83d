#!/bin/bash
# A/B of whole-step time: the default library against lib/exp_<name>.so variants, alternating, 2 rounds
# usage: bash tools/ab_step.sh <variant> [<variant> ...]
mkdir -p gpurun_out
export BENCH_SKIP_STRUCTURE_TIMING=1
for round in 1 2; do
  for v in default "$@"; do
    if [ "$v" = default ]; then unset GNNRAG_LIB; else export GNNRAG_LIB=$PWD/gnn-rag_amd/lib/exp_$v.so; fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', round(d['ms_per_step'],4), d.get('ms_per_step_fp32'))
"
  done
done
