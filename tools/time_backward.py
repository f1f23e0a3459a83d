#!/usr/bin/env python
"""Device time of the training-path operators at a bench workload (HIP events):
aggregate (unfused forward), aggregate_backward, and one full autograd layer step."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnrag_amd  # noqa: E402,F401
from gnnrag_amd import ops, stack, synth  # noqa: E402


def ev_ms(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return float(np.median(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    args = ap.parse_args()
    cfg = synth.CONFIGS[args.workload]
    dev = torch.device("cuda", 0)
    batch, feats, params = synth.make_batch(cfg), synth.make_features(cfg), synth.make_layer_params(cfg)
    devin = stack.DeviceInputs(batch, feats, dev)
    layer = stack.build_layer(cfg, batch, params, dev)
    stack.init_reason(layer, batch, devin, devin.h0)
    plan = layer.plan
    B, N, D, I = cfg.B, cfg.N, cfg.D, cfg.I
    with torch.no_grad():
        Tf = ops.linear(devin.rel_features, layer.rel_linear0.weight, layer.rel_linear0.bias)
        Ti = ops.linear(devin.rel_features_inv, layer.rel_linear0.weight, layer.rel_linear0.bias)
        dense, _ = layer(devin.seed_dist, devin.ins[0], step=0)
        g = torch.randn(B * N, 2 * I * D, device=dev)
        res = {}
        for nm, prior in (("dense", dense), ("seed", devin.seed_dist)):
            res["aggregate_%s_ms" % nm] = ev_ms(lambda: ops.aggregate(plan, prior, devin.ins[0], Tf, Ti))
            res["aggregate_backward_%s_ms" % nm] = ev_ms(
                lambda: ops.aggregate_backward(plan, prior, devin.ins[0], Tf, Ti, g))
        P = ops.relation_tables(plan, Tf, Ti, devin.ins[0], layer.e2e_linear0.weight)
        gn = torch.randn(B * N, D, device=dev)
        res["aggregate_fused_dense_ms"] = ev_ms(lambda: ops.aggregate_fused(plan, dense, P))
        res["aggregate_fused_backward_dense_ms"] = ev_ms(lambda: ops.aggregate_fused_backward(plan, dense, P, gn))
    layer.train()

    def step():
        layer.zero_grad(set_to_none=True)
        layer.local_entity_emb = devin.h0
        d = devin.seed_dist
        for j in range(cfg.L):
            d, h = layer(d, devin.ins[0], step=j)
        (d * d).sum().backward()

    with torch.enable_grad():
        for form in ("fused", "unfused"):
            layer.train_fused = form == "fused"
            res["train_fwd_bwd_%d_layers_%s_ms" % (cfg.L, form)] = ev_ms(step, 5)
    with torch.no_grad():
        layer.eval()

        def inf():
            layer.local_entity_emb = devin.h0
            d = devin.seed_dist
            for j in range(cfg.L):
                d, _ = layer(d, devin.ins[0], step=j)
        res["inference_%d_layers_ms" % cfg.L] = ev_ms(inf)
    print(res)


if __name__ == "__main__":
    main()
