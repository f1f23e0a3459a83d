#!/bin/bash
# round 5, GPU call 6: the lane-per-node light walk (k_walk_ell): structure + parity tests, A/B against k_walk_slice
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_walk_ell.py -x -q > $O/pytest_ell.log 2>&1; echo "pytest_ell rc=$?"
tail -15 $O/pytest_ell.log
cat > /tmp/ab_walk.py <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gnnrag_amd
from gnnrag_amd import ops, stack, synth
import bench
dev = torch.device("cuda", 0)
for W in sys.argv[1:]:
    cfg = synth.CONFIGS[W]
    batch = synth.make_batch(cfg); feats = synth.make_features(cfg); params = synth.make_layer_params(cfg)
    devin = stack.DeviceInputs(batch, feats, dev)
    layer = stack.build_layer(cfg, batch, params, dev)
    stack.init_reason(layer, batch, devin, devin.h0)
    with torch.no_grad():
        dense, _ = layer(devin.seed_dist, devin.ins[0], step=0)
        rl, e2e = layer.rel_linear1, layer.e2e_linear1
        Tf = ops.linear(devin.rel_features, rl.weight, rl.bias); Ti = ops.linear(devin.rel_features_inv, rl.weight, rl.bias)
        P = ops.relation_tables(layer.plan, Tf, Ti, devin.ins[0], e2e.weight)
        ms = {}
        for name, prior in (("dense", dense), ("seed", devin.seed_dist)):
            fn = lambda: ops.aggregate_fused(layer.plan, prior, P)
            fn()
            ms[name] = [round(float(np.mean(bench._events_ms(fn, 20))) * 1e3, 1) for _ in range(3)]
    print("AB", W, os.environ.get("GNNRAG_WALK_ELL", "default(1)"), ms)
PY
for r in 1 2; do
  GNNRAG_WALK_ELL=1 timeout 300 python /tmp/ab_walk.py C2 C2u C4 2>&1 | grep AB
  GNNRAG_WALK_ELL=0 timeout 300 python /tmp/ab_walk.py C2 C2u C4 2>&1 | grep AB
done | tee $O/ab_walk.log
for i in 1 2; do
  GNNRAG_WALK_ELL=1 timeout 400 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_ell_$i.json 2> $O/bench_ell_$i.err
  GNNRAG_WALK_ELL=0 timeout 400 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_old_$i.json 2> $O/bench_old_$i.err
done
for f in $O/bench_ell_1.json $O/bench_old_1.json $O/bench_ell_2.json $O/bench_old_2.json; do tail -1 $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', round(d['ms_per_step'],4), round(d['roofline']['frac'],3), {k:round(v,4) for k,v in d.get('kernel_ms',{}).items() if 'fused' in k or 'csr' in k}, round(d['csr_build_ms'],3))"; done
