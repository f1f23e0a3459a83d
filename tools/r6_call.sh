cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
for W in C1 C3; do
for i in 1 2; do
python bench.py --workload $W --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$W', round(d['ms_per_step'],4), round(d['step_ms_spread']['p50'],4), round(d['step_ms_spread']['max'],3))
"
done
done 2>&1 | tee gpurun_out/r06k/c1_c3_gc_off.txt
