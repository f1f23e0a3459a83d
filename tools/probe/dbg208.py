import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import gnnrag_amd
from gnnrag_amd import ops, stack, synth
dev = torch.device("cuda", 0)
B, N, E, R, D, I = 5, 1800, 9000, 400, int(os.environ.get("DD", "208")), 2
cfg = synth.GraphConfig(name="sweep", B=B, N=N, E=E, R=R, D=D, I=I, L=2, T=2, seed=B * 1000 + D,
                        normalized_gnn=(D % 2 == 0), pos_emb=(I % 2 == 1), n_real_min=max(2, N // 2))
batch = synth.make_batch(cfg); feats = synth.make_features(cfg); params = synth.make_layer_params(cfg)
for mode in (ops.MATH_MIXED, ops.MATH_FP32):
    ops.set_dense_math(mode)
    g1 = stack.run_stack(batch, feats, params, dev, use_type_layer=True, norm_rel=True, path=1)
    g2 = stack.run_stack(batch, feats, params, dev, use_type_layer=True, norm_rel=True, path=2)
    a, b = g1["h"][0].reshape(B * N, D), g2["h"][0].reshape(B * N, D)
    bad = np.argwhere(np.abs(a - b) > 1e-3 * max(1, np.abs(a).max()))
    print("math", mode, "bad", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:20], "cols", sorted(set(bad[:, 1].tolist()))[:40])
    if len(bad):
        r, c = bad[0]
        print(" first", r, c, a[r, c], b[r, c], "row nonzero cols unfused/fused:", (a[r] != 0).sum(), (b[r] != 0).sum())
