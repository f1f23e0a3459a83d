cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; shift; env "$@" > gpurun_out/c3x.log 2>&1; tail -1 gpurun_out/c3x.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['step_ms_spread'] and round(d['step_ms_spread']['p50'],4), d['step_ms_spread'] and round(d['step_ms_spread']['max'],2))"; }
run "C3 default" python bench.py --workload C3 --no-cpu-baseline --steps 30 --spread-steps 32 --fp32-steps 0
run "C3 skip structure timing" BENCH_SKIP_STRUCTURE_TIMING=1 python bench.py --workload C3 --no-cpu-baseline --steps 30 --spread-steps 32 --fp32-steps 0
run "C3 warmup 100" python bench.py --workload C3 --no-cpu-baseline --steps 30 --warmup 100 --spread-steps 32 --fp32-steps 0
run "C3 warmup 5 steps 20" python bench.py --workload C3 --no-cpu-baseline --steps 20 --warmup 5 --spread-steps 32 --fp32-steps 0
run "C2 warmup 5 steps 20" python bench.py --no-cpu-baseline --steps 20 --warmup 5 --spread-steps 64 --fp32-steps 0
run "C2 warmup 5 steps 20 again" python bench.py --no-cpu-baseline --steps 20 --warmup 5 --spread-steps 64 --fp32-steps 0
