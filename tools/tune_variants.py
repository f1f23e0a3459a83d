#!/usr/bin/env python
"""A/B harness for compile-time kernel variants: builds lib/exp_<name>.so per variant (CPU side,
`--build`), then (GPU side, `--run`) times the hot ops of workload C2 with each variant in its own
process and prints one line per variant.
    python tools/tune_variants.py --build          # here (hipcc cross-compiles)
    python tools/tune_variants.py --run            # on the GPU box (gpurun)
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

VARIANTS = {
    "full": {"GNNRAG_SLICE_ABLATE": 0},
    "nomed": {"GNNRAG_SLICE_ABLATE": 64},
    "nohuge": {"GNNRAG_SLICE_ABLATE": 128},
    "nopass2": {"GNNRAG_SLICE_ABLATE": 2},
    "nowalk": {"GNNRAG_SLICE_ABLATE": 2 + 32 + 16},
}

CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import gnnrag_amd
from gnnrag_amd import ops, stack, synth
import bench
dev = torch.device("cuda", 0)
cfg = synth.CONFIGS["C2"]
batch = synth.make_batch(cfg); feats = synth.make_features(cfg); params = synth.make_layer_params(cfg)
devin = stack.DeviceInputs(batch, feats, dev)
layer = stack.build_layer(cfg, batch, params, dev)
stack.init_reason(layer, batch, devin, devin.h0)
with torch.no_grad():
    dense, _ = layer(devin.seed_dist, devin.ins[0], step=0)
    rl, e2e = layer.rel_linear1, layer.e2e_linear1
    Tf = ops.linear(devin.rel_features, rl.weight, rl.bias); Ti = ops.linear(devin.rel_features_inv, rl.weight, rl.bias)
    P = ops.relation_tables(Tf, Ti, devin.ins[0], e2e.weight)
    ms = {}
    for name, prior in (("fused_dense", dense), ("fused_seed", devin.seed_dist)):
        ops.aggregate_fused(layer.plan, prior, P)
        ms[name] = float(np.mean(bench._events_ms(lambda: ops.aggregate_fused(layer.plan, prior, P), 10)))
print("RESULT " + json.dumps(ms))
'''


def main():
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import build
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(VARIANTS)
    if "--build" in sys.argv:
        for n in names:
            print(build.build_variant(n, VARIANTS[n]))
    if "--run" in sys.argv:
        for n in names:
            env = dict(os.environ, GNNRAG_LIB=os.path.join(build.LIBDIR, "exp_%s.so" % n))
            r = subprocess.run([sys.executable, "-c", CHILD % REPO], env=env, capture_output=True, text=True,
                               timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(n, "FAILED", r.stderr[-800:])
                continue
            ms = json.loads(line[0][7:])
            print("%-10s " % n + "  ".join("%s=%.1f" % (k[:18], v * 1e3) for k, v in ms.items()))


if __name__ == "__main__":
    main()
