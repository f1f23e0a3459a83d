#!/usr/bin/env python
"""Where does a step's HOST time go?  Per-step host enqueue times of the eager step (no sync inside the loop), for one
workload: a one-off multi-millisecond stall of the HIP runtime shows up as a single outlier, a uniformly slow host path
as a high median.  python tools/probe/host_stall_probe.py C3 [steps]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import gnnrag_amd  # noqa: E402,F401
from gnnrag_amd import stack, synth  # noqa: E402


def main():
    w = sys.argv[1] if len(sys.argv) > 1 else "C3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    sync_every = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cfg = synth.CONFIGS[w]
    dev = torch.device("cuda", 0)
    batch, feats, params = synth.make_batch(cfg), synth.make_features(cfg), synth.make_layer_params(cfg)
    devin = stack.DeviceInputs(batch, feats, dev)
    layer = stack.build_layer(cfg, batch, params, dev)
    stack.init_reason(layer, batch, devin, devin.h0)
    ts = []
    with torch.no_grad():
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for i in range(n):
            t0 = time.perf_counter()
            layer.local_entity_emb = devin.h0
            stack.run_layers(layer, cfg, devin)
            ts.append((time.perf_counter() - t0) * 1e3)
            if sync_every and (i + 1) % sync_every == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t_all) * 1e3 / n
    ts = np.array(ts)
    top = np.argsort(-ts)[:6]
    print("%s: wall %.3f ms/step over %d steps (sync every %d); host enqueue p50 %.3f p95 %.3f max %.3f; largest: %s"
          % (w, wall, n, sync_every, np.percentile(ts, 50), np.percentile(ts, 95), ts.max(),
             [(int(i), round(float(ts[i]), 2)) for i in top]))


if __name__ == "__main__":
    main()
