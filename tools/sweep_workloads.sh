cd $GRAFT_REPO_ROOT
for W in C1 C3 C4 C5 C2fb; do
  python bench.py --workload $W --no-cpu-baseline --steps 30 --spread-steps 0 --fp32-steps 0 > gpurun_out/sw_$W.log 2>&1
  tail -1 gpurun_out/sw_$W.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$W', round(d['ms_per_step'],4), r['kernel'][:70], round(r['frac'],3), {k:round(v,4) for k,v in d['kernel_ms'].items()})"
done
for W in C1 C3; do
  python bench.py --workload $W --no-cpu-baseline --steps 30 --spread-steps 0 --fp32-steps 0 --launch graph > gpurun_out/sw_${W}_graph.log 2>&1
  tail -1 gpurun_out/sw_${W}_graph.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$W graph', round(d['ms_per_step'],4))"
done
