import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import gnnrag_amd
from gnnrag_amd import stack, synth
dev = torch.device("cuda", 0)
cfg = synth.CONFIGS["C2"]
batch, feats, params = synth.make_batch(cfg), synth.make_features(cfg), synth.make_layer_params(cfg)
devin = stack.DeviceInputs(batch, feats, dev)
layer = stack.build_layer(cfg, batch, params, dev)
stack.init_reason(layer, batch, devin, devin.h0)
def step():
    layer.local_entity_emb = devin.h0
    d, _ = stack.run_layers(layer, cfg, devin)
    return d
with torch.no_grad():
    for _ in range(3): step()
    torch.cuda.synchronize()
    n = 300
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); step(); b.record()
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) for a, b in evs])
    print("first 30:", np.round(ts[:30], 3).tolist())
    print("outliers >1.0:", [(i, round(float(t), 2)) for i, t in enumerate(ts) if t > 1.0])
    print("mean first 20 %.4f, mean 20..40 %.4f, p50 %.4f" % (ts[:20].mean(), ts[20:40].mean(), np.median(ts)))
    print("alloc stats:", torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["segment.all.allocated"])
