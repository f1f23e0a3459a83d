#!/usr/bin/env python
"""Where in a process's life do the HIP runtime's one-off host stalls fall?  Enqueues N small launches (alternating a
library kernel and a torch op), records the host time of every enqueue, synchronises every 32, and prints every enqueue
that took longer than 2 ms with its launch index - bench.py's timed loops of the many-launch workloads (C1 / C3: ~45
launches per step) caught one such stall of ~35 ms inside a 30-step region (profiles/r06j_other_workloads.json: 1.48 /
1.69 ms per step in the loop, 0.34 / 0.48 in the 200 single-step measurements behind it)."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402
import gnnrag_amd  # noqa: E402,F401
from gnnrag_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
x = torch.randn(4, 2000, device=dev)
y = torch.empty_like(x)
ops.masked_softmax(x.reshape(-1), 4, 2000)
torch.cuda.synchronize()
slow = []
t_all = time.perf_counter()
for i in range(n):
    t0 = time.perf_counter()
    if i & 1:
        ops.masked_softmax(x.reshape(-1), 4, 2000)
    else:
        torch.add(x, 1.0, out=y)
    dt = time.perf_counter() - t0
    if dt > 2e-3:
        slow.append((i, round(dt * 1e3, 2)))
    if i % 32 == 31:
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt > 2e-3:
            slow.append((i, "sync %.2f" % (dt * 1e3)))
print("launches %d, wall %.1f ms, enqueues / syncs above 2 ms (index, ms): %s" % (n, (time.perf_counter() - t_all) * 1e3, slow))
