#!/bin/bash
# round 5, GPU call 9: host profile of a steady-state forward (d200 batch 64 / 16, d50 batch 1) + the host cost of F.linear
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 bash tools/profile_forward.sh d200 64 torch
timeout 300 bash tools/profile_forward.sh d200 64 cprofile
timeout 300 bash tools/profile_forward.sh d50 1 torch
timeout 300 bash tools/profile_forward.sh d50 1 cprofile
timeout 200 python - <<'PY' > gpurun_out/fwd_prof/linear_host_cost.txt 2>&1
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import gnnrag_amd
from gnnrag_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def host_us(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("shape (M, K -> N)            F.linear host / wall us     ops.linear host / wall us     matmul+add host / wall")
with torch.no_grad():
    for M, K, N in ((64, 200, 200), (64, 400, 200), (64 * 12, 200, 200), (600, 768, 200), (64, 200, 1), (128000, 200, 200), (1, 50, 50), (12, 50, 50)):
        x = torch.randn(M, K, device=dev); lin = torch.nn.Linear(K, N).to(dev)
        a = host_us(lambda: torch.nn.functional.linear(x, lin.weight, lin.bias))
        b = host_us(lambda: ops.linear(x, lin.weight, lin.bias))
        wt = lin.weight.t().contiguous()
        c = host_us(lambda: torch.addmm(lin.bias, x, wt))
        print("%-28s %8.1f / %8.1f          %8.1f / %8.1f          %8.1f / %8.1f" % ((M, K, N), a[0], a[1], b[0], b[1], c[0], c[1]))
    x = torch.randn(64, 2000, device=dev)
    for name, fn in (("softmax", lambda: torch.softmax(x, 1)), ("add", lambda: x + x), ("cat", lambda: torch.cat((x, x), 1)),
                     ("max", lambda: torch.max(x, 1)), ("empty", lambda: torch.empty(64, 2000, device=dev))):
        a = host_us(fn)
        print("%-28s %8.1f / %8.1f" % (name, a[0], a[1]))
PY
cat gpurun_out/fwd_prof/linear_host_cost.txt
