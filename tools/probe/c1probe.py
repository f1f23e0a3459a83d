import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import gnnrag_amd
from gnnrag_amd import ops, stack, synth
dev = torch.device("cuda", 0)
cfg = synth.CONFIGS["C1"]
batch = synth.make_batch(cfg); feats = synth.make_features(cfg); params = synth.make_layer_params(cfg)
devin = stack.DeviceInputs(batch, feats, dev)
layer = stack.build_layer(cfg, batch, params, dev)
stack.init_reason(layer, batch, devin, devin.h0)
with torch.no_grad():
    layer.local_entity_emb = devin.h0
    stack.run_layers(layer, cfg, devin)
    st = layer._stack
    torch.cuda.synchronize()
    for name, fn in (("stack.run", lambda: st.run(devin.h0, devin.seed_dist, devin.ins[0])),
                     ("softmax", lambda: ops.masked_softmax(devin.seed_dist.reshape(-1), cfg.B, cfg.N)),
                     ("linear", lambda: ops.linear(devin.rel_features, layer.rel_linear0.weight, layer.rel_linear0.bias))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s: host enqueue %.1f us/call, total %.1f us/call" % (name, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
