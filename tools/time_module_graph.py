#!/usr/bin/env python
"""What hipGraph replay buys THROUGH THE MODULE (the path main.py takes), BASELINE config 1 (one WebQSP-shaped question,
D = 50, 3 iterations x 3 layers): per forward = init_reason (new batch -> new stack, so a capture per forward when the
replay is on) + num_iter x num_gnn ReasonGNNLayer.forward calls.  Eager vs GNNRAG_GRAPH=1, wall clock over many forwards."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import gnnrag_amd  # noqa: E402,F401
from gnnrag_amd import stack, synth  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C1"]
    batch, feats, params = synth.make_batch(cfg), synth.make_features(cfg), synth.make_layer_params(cfg)
    dvi = stack.DeviceInputs(batch, feats, dev)
    out = {}
    for mode in ("eager", "graph", "eager", "graph"):
        layer = stack.build_layer(cfg, batch, params, dev)
        layer.graph_small = mode == "graph"

        def forward():
            stack.init_reason(layer, batch, dvi, dvi.h0)          # a new batch every forward (plan_for caches the structure)
            d, _ = stack.run_layers(layer, cfg, dvi)
            return d
        with torch.no_grad():
            for _ in range(20):
                forward()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                forward()
            torch.cuda.synchronize()
            out.setdefault(mode, []).append((time.perf_counter() - t0) * 1e3 / n)
    print("module path, %s (B=%d, N=%d, D=%d, T=%d, L=%d): ms per forward  eager %s  graph (capture per forward + %d replays) %s"
          % (cfg.name, cfg.B, cfg.N, cfg.D, cfg.T, cfg.L, ["%.3f" % x for x in out["eager"]], cfg.T - 1,
             ["%.3f" % x for x in out["graph"]]))


if __name__ == "__main__":
    main()
