#!/usr/bin/env python
"""Host (CPU) cost of enqueueing one ReasonGNNLayer.forward vs the device time of its kernels, at a small
workload where the path is launch bound (C1: one question)."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnrag_amd  # noqa: E402,F401
from gnnrag_amd import stack, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C1")
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
cfg = synth.CONFIGS[args.workload]
dev = torch.device("cuda", 0)
batch, feats, params = synth.make_batch(cfg), synth.make_features(cfg), synth.make_layer_params(cfg)
devin = stack.DeviceInputs(batch, feats, dev)
layer = stack.build_layer(cfg, batch, params, dev)
stack.init_reason(layer, batch, devin, devin.h0)


def step():
    d = devin.seed_dist
    for j in range(cfg.L):
        d, _ = layer(d, devin.ins[0], step=j)
    return d


with torch.no_grad():
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print({"workload": cfg.name, "enqueue_us_per_layer_call": t_enq / n / cfg.L * 1e6,
           "wall_us_per_layer_call": t_all / n / cfg.L * 1e6})
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
