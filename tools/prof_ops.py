#!/usr/bin/env python
"""Runs the hot-path ops of one workload a few times each, op by op, so that rocprofv3
kernel traces / PMC counters are attributable per kernel.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o name -- python tools/prof_ops.py --workload C2 --reps 5
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gnnrag_amd  # noqa: E402
from gnnrag_amd import ops, stack, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ops", default="rel,agg,upd,tab,aggf,updf,sm,layer")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = synth.CONFIGS[a.workload]
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    devin = stack.DeviceInputs(batch, feats, dev)
    layer = stack.build_layer(cfg, batch, params, dev)
    stack.init_reason(layer, batch, devin, devin.h0)
    B, N, D, I = cfg.B, cfg.N, cfg.D, cfg.I
    which = a.ops.split(",")
    with torch.no_grad():
        dist1, _ = layer(devin.seed_dist, devin.ins[0], step=0)          # a dense prior
        h = devin.h0.reshape(B * N, D)
        rl, e2e = layer.rel_linear1, layer.e2e_linear1
        Tf = ops.linear(devin.rel_features, rl.weight, rl.bias)
        Ti = ops.linear(devin.rel_features_inv, rl.weight, rl.bias)
        agg = ops.aggregate(layer.plan, dist1, devin.ins[0], Tf, Ti)
        nbr0 = agg[:, :D].contiguous()                                          # a dense [BN, D] stand-in for nbr
        torch.cuda.synchronize()
        for _ in range(a.reps):
            if "rel" in which:
                ops.linear(devin.rel_features, rl.weight, rl.bias)
            if "agg" in which:
                ops.aggregate(layer.plan, dist1, devin.ins[0], Tf, Ti)          # dense prior
                ops.aggregate(layer.plan, devin.seed_dist, devin.ins[0], Tf, Ti)  # sparse (seed) prior
            if "upd" in which:
                h2, sc = ops.update_score(h, agg, e2e.weight, e2e.bias, layer.score_func.weight,
                                          layer.score_func.bias, layer.local_entity_mask, I)
                if "sm" in which:
                    ops.masked_softmax(sc, B, N)
            if "tab" in which:
                P = ops.relation_tables(layer.plan, Tf, Ti, devin.ins[0], e2e.weight)
            if "aggfd" in which:                                                # fused walk, dense prior only (what a step runs)
                Pd = ops.relation_tables(layer.plan, Tf, Ti, devin.ins[0], e2e.weight)
                ops.aggregate_fused(layer.plan, dist1, Pd)
            if "aggf" in which:
                P = ops.relation_tables(layer.plan, Tf, Ti, devin.ins[0], e2e.weight) if "tab" not in which else P
                nbr = ops.aggregate_fused(layer.plan, dist1, P)                 # dense prior
                ops.aggregate_fused(layer.plan, devin.seed_dist, P)              # sparse (seed) prior
                if "updf" in which:
                    ops.update_score_fused(h, nbr, e2e.weight, e2e.bias, layer.score_func.weight,
                                           layer.score_func.bias, layer.local_entity_mask, I)
            if "updfd" in which:                                                # the self-block update alone (dense nbr)
                ops.update_score_fused(h, nbr0, e2e.weight, e2e.bias,
                                       layer.score_func.weight, layer.score_func.bias, layer.local_entity_mask, I)
            if "fr" in which:                                                   # seed-prior (frontier) form, piece by piece
                fr = ops.Frontier(layer.plan, devin.seed_dist)
                Pf = fr.relation_tables(Tf, Ti, devin.ins[0], e2e.weight)
                fr.aggregate(Pf)
            if "layer0" in which:                                               # a whole layer from the seed prior
                layer.local_entity_emb = devin.h0
                layer(devin.seed_dist, devin.ins[0], step=0)
            if "layer" in which:
                layer.local_entity_emb = devin.h0
                layer(dist1, devin.ins[0], step=1)
        torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
