#!/usr/bin/env python
"""Host wall-clock of the Evaluator's candidate selection for one C2-sized batch (64 x 2000 slots):
the reference's per-slot Python loop (restated in oracle/eval_tail.py; evaluate.py:188-207 + :34-51)
vs gnnrag_amd.eval_tail.retrieved_candidates (one kernel + one small D2H)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnrag_amd  # noqa: E402,F401
import oracle.eval_tail as oe  # noqa: E402
from gnnrag_amd import eval_tail  # noqa: E402

B, N, eps = 64, 2000, 0.95
rng = np.random.default_rng(0)
logits = rng.standard_normal((B, N)) * 4
p = np.exp(logits - logits.max(1, keepdims=True))
p = (p / p.sum(1, keepdims=True)).astype(np.float32)
cands = rng.integers(0, 10 ** 5, size=(B, N))
seeds = np.zeros((B, N))
seeds[:, 0] = 1.0
pad, ignore = 10 ** 6, (1 - eps) / N
dev = torch.device("cuda", 0)
pd = torch.from_numpy(p).to(dev)
eval_tail.retrieved_candidates(pd, cands, seeds, pad, ignore, eps)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    got = eval_tail.retrieved_candidates(pd, cands, seeds, pad, ignore, eps)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
pc = pd.cpu()
want = []
for b in range(B):
    kept, cut = oe.select(pc[b].tolist(), cands[b].tolist(), seeds[b].tolist(), pad, ignore, eps)
    want.append(([(int(cands[b, j]), float(p[b, j])) for j in kept[:cut]], len(kept)))
t_ref = time.perf_counter() - t0
assert got == want
print({"python_loop_ms": t_ref * 1e3, "device_ms": t_dev * 1e3, "mean_retrieved": float(np.mean([len(x[0]) for x in got]))})
