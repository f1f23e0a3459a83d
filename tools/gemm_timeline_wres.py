#!/usr/bin/env python
"""Per-wave phase stamps of k_gemm_wres (timing build): start, W staged, then per tile (start, k loop done), end."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
import gnnrag_amd  # noqa
from gnnrag_amd import _lib, ops, stack, synth
dev = torch.device("cuda", 0)
lib = _lib.load()
fn = lib.gnnrag_debug_set_timing_buffer
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
cfg = synth.CONFIGS["C2"]
batch = synth.make_batch(cfg); feats = synth.make_features(cfg); params = synth.make_layer_params(cfg)
devin = stack.DeviceInputs(batch, feats, dev)
layer = stack.build_layer(cfg, batch, params, dev)
stack.init_reason(layer, batch, devin, devin.h0)
B, N, D, I = cfg.B, cfg.N, cfg.D, cfg.I
e2e, sf = layer.e2e_linear1, layer.score_func
h = devin.h0.reshape(B * N, D)
nbr = torch.randn_like(h)
f = lambda: ops.update_score_fused(h, nbr, e2e.weight, e2e.bias, sf.weight, sf.bias, layer.local_entity_mask, I, math=0)
f(); torch.cuda.synchronize()
tbuf = torch.zeros((4096, 32), dtype=torch.int64, device=dev)
assert fn(tbuf.data_ptr()) == 0
f(); torch.cuda.synchronize()
fn(None)
t = tbuf.cpu().numpy()
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
us = lambda x: x / 2250.0
print("waves", len(t), "span us", us(t[:, :30].max() - t0))
print("start spread", us(np.percentile(t[:, 0] - t0, [0, 50, 100])))
print("W staged (since start)", us(np.percentile(t[:, 1] - t[:, 0], [0, 50, 100])))
nt = ((t[:, 2:28] != 0).sum(1)) // 2
print("tiles per wave", np.bincount(nt))
for k in range(int(nt.max())):
    ok = nt > k
    a, b = t[ok, 2 + 2 * k], t[ok, 3 + 2 * k]
    nxt = np.where(nt[ok] > k + 1, t[ok, 4 + 2 * k], t[ok, 2 + 2 * nt[ok]])
    print("tile %d: start %s kloop %s epilogue %s" % (k, us(np.percentile(a - t0, [0, 50, 100])).round(1),
          us(np.percentile(b - a, [0, 50, 100])).round(1), us(np.percentile(nxt - b, [0, 50, 100])).round(1)))
end = np.array([t[i, 2 + 2 * nt[i]] for i in range(len(t))])
print("copy (entry -> before barrier)", us(np.percentile(t[:, 0] - t[:, 31], [0, 50, 100])).round(1))
print("entry -> end per wave", us(np.percentile(end - t[:, 31], [0, 50, 100])).round(1))
print("entry -> first tile start", us(np.percentile(t[:, 2] - t[:, 31], [0, 50, 100])).round(1))
