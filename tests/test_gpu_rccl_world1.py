"""The multi-GPU path on the ONE GPU the driver's test box has ("Next round" item 9): a process group
over RCCL (backend "nccl") with world size 1, through the same code every rank runs at world size 8 -

* ``shard.gather_rows_async`` / ``gather_rows``: ``all_gather_into_tensor`` enqueued behind a batch's kernels and waited
  for one batch later (the overlap of DESIGN section 6), result identical to the local shard;
* ``shard.shard_model`` under GNNRAG_FORCE_DIST=1: question ranges, the gather of the scored nodes and the all-reduce of
  the loss around a model function;
* ``bench.py`` with GNNRAG_FORCE_DIST=1 in a subprocess: rendezvous on 127.0.0.1, barrier-bracketed timing, MAX over
  ranks, one JSON line.
World sizes > 1 are covered on CPU over gloo (tests/test_shard_gloo.py); the unmodified main.py under the same forced
group is tests/test_gpu_main_py.py.  Reference: SURVEY.md section 8e."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def rccl_world1():
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_async_gather_of_scored_nodes_over_rccl(rccl_world1):
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import shard
    dev = torch.device("cuda", 0)
    B, N = 7, 1234
    g = torch.Generator().manual_seed(3)
    d1 = torch.rand(B, N, generator=g).to(dev)
    fin1 = shard.gather_rows_async(d1, B, ranges=[(0, B)])
    d2 = (d1 * 2 + 1).contiguous()                       # "the next batch's kernels" enqueued behind the collective
    fin2 = shard.gather_rows_async(d2, B)
    assert torch.equal(fin1(), d1) and torch.equal(fin2(), d2)
    assert torch.equal(shard.gather_rows(d1[:, :5].contiguous(), B), d1[:, :5])
    with pytest.raises(ValueError):
        shard.gather_rows_async(d1[:3], B)               # a shard that does not match this rank's range


def test_shard_model_runs_both_collectives_at_world_size_1(rccl_world1, monkeypatch):
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import shard
    dev = torch.device("cuda", 0)
    B, N = 5, 64
    calls = []

    class Model:
        def forward(self, batch, training=False):
            calls.append(batch[0].shape[0])
            dist_ = torch.softmax(torch.from_numpy(batch[4]).float().to(dev) * 3, dim=1)
            return dist_.sum() * 0 + 1.25, dist_.argmax(1), dist_, None

    rng = np.random.default_rng(0)
    bids = np.repeat(np.arange(B), 4)
    et = (bids * N, np.zeros(len(bids), np.int64), bids * N + 1, bids, np.arange(len(bids)), [1.0] * len(bids), [1.0] * len(bids))
    batch = (np.zeros((B, N), np.int64), np.zeros((B, N)), et, np.zeros((B, 3), np.int64), rng.random((B, N)), None,
             np.zeros((B, N)))
    m = shard.shard_model(Model())
    monkeypatch.setenv("GNNRAG_FORCE_DIST", "1")
    loss, pred, full, _ = m.forward(batch)
    want = torch.softmax(torch.from_numpy(batch[4]).float().to(dev) * 3, dim=1)
    assert calls == [B] and torch.equal(full, want) and torch.equal(pred, want.argmax(1))
    assert abs(float(loss) - 1.25) < 1e-6                # all-reduced batch mean of one rank's loss


def test_bench_under_forced_rccl_group():
    env = dict(os.environ, GNNRAG_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               BENCH_SKIP_STRUCTURE_TIMING="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "C1", "--steps", "5", "--warmup", "3",
                        "--no-cpu-baseline", "--spread-steps", "0", "--fp32-steps", "0"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["ms_per_step"] > 0 and d["scaling"] == "weak"
