import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import gnnrag_amd
from gnnrag_amd import ops, stack, synth
dev = torch.device("cuda", 0)
torch.manual_seed(0)
D = 208
for M in (9000, 8192, 16000, 9008):
    h = torch.randn(M, D, device=dev); nbr = torch.randn(M, D, device=dev)
    W = torch.randn(D, 5 * D, device=dev) / 14; b = torch.randn(D, device=dev)
    ws = torch.randn(D, device=dev); bs = torch.randn(1, device=dev); mask = torch.ones(M, device=dev)
    o0 = ops.update_score_fused(h, nbr, W, b, ws, bs, mask, 2, math=ops.MATH_FP32)
    o1 = ops.update_score_fused(h, nbr, W, b, ws, bs, mask, 2, math=ops.MATH_BF16X3)
    d = (o0[0] - o1[0]).abs()
    bad = (d > 1e-3).nonzero()
    print("update M", M, "bad", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:8], "cols", sorted(set(bad[:, 1].tolist()))[:8],
          "score diff", float((o0[1] - o1[1]).abs().max()))
