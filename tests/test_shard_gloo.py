"""Multi-process (world_size 2, gloo, CPU) test of the question-sharded layout:
shard -> per-rank compute -> one all-gather == unsharded result, bit for bit.
The per-rank compute is the CPU oracle here (no GPU in this container); on GPUs the same
shard/gather code runs over RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q, balance="count"):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gnnrag_amd  # noqa: F401
        from gnnrag_amd import shard, synth
        import oracle.rearev_torch_cpu as otorch
        torch.set_num_threads(1)
        cfg = synth.GraphConfig(name="s", B=B, N=24, E=60, R=5, D=16, I=2, L=2, T=1, seed=9, n_real_min=3)
        batch = synth.make_batch(cfg)
        feats = synth.make_features(cfg)
        params = synth.make_layer_params(cfg)
        full = otorch.run_stack(batch, feats, params)["dist"][-1]
        ref_batch = (batch.local_entity, batch.query_entities, batch.edge_tuple, np.zeros((B, 1)),
                     batch.seed_dist, None, np.zeros((B, cfg.N)))
        ranges = shard.shard_ranges(ref_batch, world, balance)
        lo, hi = ranges[rank]
        sb = shard.shard_batch(ref_batch, rank, world, balance)
        sub = synth.Batch(cfg=synth.GraphConfig(**{**cfg.__dict__, "B": hi - lo}), local_entity=sb[0],
                          query_entities=sb[1], seed_dist=sb[4], edge_tuple=sb[2],
                          num_entity=batch.num_entity, n_real=batch.n_real[lo:hi])
        sfe = dict(feats)
        sfe["h0"] = feats["h0"][lo:hi]
        sfe["ins"] = feats["ins"][:, lo:hi]
        local = otorch.run_stack(sub, sfe, params)["dist"][-1]
        gathered = shard.gather_rows(torch.from_numpy(local), B, ranges=None if balance == "count" else ranges).numpy()
        ok = gathered.shape == full.shape and np.array_equal(gathered, full)
        q.put((rank, bool(ok), float(np.abs(gathered - full).max()) if gathered.shape == full.shape else -1.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B,balance", [(6, "count"), (5, "count"), (7, "facts")])   # even / ragged / fact-balanced split
def test_shard_and_gather_world2(B, balance):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q, balance)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err in res:
        assert ok, "rank %d: gathered != unsharded (max err %g)" % (rank, err)


def test_question_range_partition():
    from gnnrag_amd import shard
    for B in (1, 5, 64, 250):
        for world in (1, 2, 3, 8):
            rs = [shard.question_range(B, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in rs]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_ranges_properties():
    """Fact-balanced contiguous split: a partition of [0, B), every rank non-empty while
    questions remain, never worse than the even-count split by more than the largest question."""
    from gnnrag_amd import shard
    rng = np.random.default_rng(3)
    for _ in range(500):
        B, world = int(rng.integers(1, 60)), int(rng.integers(1, 9))
        w = rng.integers(0, 3000, B) * (rng.random(B) < 0.8)
        r = shard.balanced_ranges(w, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == B
        assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(r, r[1:]))
        if B >= world:
            assert all(h > l for l, h in r)
        load = max(int(w[l:h].sum()) for l, h in r)
        even = max(int(w[l:h].sum()) for l, h in (shard.question_range(B, k, world) for k in range(world)))
        assert load <= even + int(w.max()), (w, world, r)
    # WebQSP-like: a few huge subgraphs among many small ones
    w = np.array([40, 35, 9000, 60, 20, 7000, 30, 45, 50, 8000, 25, 30])
    r = shard.balanced_ranges(w, 4)
    assert max(w[l:h].sum() for l, h in r) == 9000 + 75 or max(w[l:h].sum() for l, h in r) <= 9200


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import contextlib
    import io
    import json
    import bench
    buf = io.StringIO()
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--dry-run-cpu", "--scaling", "strong"]
    with contextlib.redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
    q.put((rank, json.loads(lines[-1]) if lines else None))


def test_bench_distributed_skeleton_world2():
    """bench.py's multi-rank skeleton without GPUs (--dry-run-cpu: gloo, the CPU restatement as the per-rank step):
    rendezvous from the launcher's environment, strong-scaling split of ONE global batch by facts, the all-gather,
    the barrier-bracketed timing and its MAX over ranks, and exactly one JSON line from rank 0."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None and res[0] is not None
    out = res[0]
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "strong" and out["value"] > 0
    assert out["gathered_matches_unsharded"] is True


def test_bench_self_launches_its_ranks_without_a_launcher():
    """`python bench.py --gpus 2` as the driver may call it - no torchrun, WORLD_SIZE unset: bench.py starts its two
    ranks itself (torch.distributed.run on 127.0.0.1), rank 0 prints the ONE JSON line, the return code is the ranks'."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--dry-run-cpu", "--scaling", "strong"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["gathered_matches_unsharded"] is True
    # a failing rank must fail the launcher (rc propagates)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--dry-run-cpu", "--workload", "no-such-workload-for-rc"], env=dict(env, GNNRAG_DRYRUN_FAIL="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


REF = "/root/reference/gnn"


def _model_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import test_dropin_with_reference as td
        from gnnrag_amd import shard
        args, dataset, model = td.build_reference_setup()       # same seeds in every process: same model, same data
        test = dataset["test"]
        test.reset_batches(is_sequential=True)
        np.random.seed(11)
        batch = test.get_batch(0, 5, fact_dropout=0.0, test=True)
        with torch.no_grad():
            loss_ref, pred_ref, dist_ref, _ = model(batch[:-1])
            shard.shard_model(model)
            loss, pred, full, _ = model(batch[:-1])
        ok = (full.shape == dist_ref.shape and float((full - dist_ref).abs().max()) <= 1e-6 and
              torch.equal(pred, pred_ref) and abs(float(loss) - float(loss_ref)) <= 1e-5 * max(1.0, abs(float(loss_ref))))
        q.put((rank, bool(ok), float((full - dist_ref).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(not os.path.isdir(REF), reason="live reference not available")
def test_sharded_reference_model_world2():
    """The live reference ReaRev (CPU) behind shard.shard_model on 2 gloo ranks with a ragged split (5 questions):
    gathered pred_dist, pred and the batch-mean loss equal the unsharded forward on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err in res:
        assert ok, "rank %d: sharded forward differs (max err %g)" % (rank, err)


def _rank_local_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import test_dropin_with_reference as td
        from gnnrag_amd import shard
        from gnnrag_amd.data import fact_mat
        args, dataset, model = td.build_reference_setup()
        test = dataset["test"]
        test.reset_batches(is_sequential=True)
        fact_mat.patch_loader(test, cache=True)
        full_batch = test.get_batch(0, 5, fact_dropout=0.0, test=True)
        with torch.no_grad():
            loss_ref, pred_ref, dist_ref, _ = model(full_batch[:-1])
        fact_mat.patch_loader(test, cache=True, shard=(rank, world))
        batch = test.get_batch(0, 5, fact_dropout=0.0, test=True)
        sf = batch[2]
        lo, hi = sf.ranges[rank]
        built = len(sf.local[0])
        want = int(shard.facts_per_question(full_batch[2], 5)[lo:hi].sum())
        try:
            sf[0]
            refused = False
        except TypeError:
            refused = True
        with torch.no_grad():
            shard.shard_model(model)
            loss, pred, full, _ = model(batch[:-1])
        ok = (isinstance(sf, fact_mat.ShardedFacts) and built == want and refused
              and np.array_equal(sf.facts_per_question, shard.facts_per_question(full_batch[2], 5))
              and float((full - dist_ref).abs().max()) <= 1e-6 and torch.equal(pred, pred_ref)
              and abs(float(loss) - float(loss_ref)) <= 1e-5 * max(1.0, abs(float(loss_ref))))
        q.put((rank, bool(ok), float((full - dist_ref).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(not os.path.isdir(REF), reason="live reference not available")
def test_rank_builds_only_its_own_questions_world2():
    """`patch_loader(..., shard=(rank, world))`: every rank builds the tuple of ITS fact-balanced question range only
    (`fact_mat.ShardedFacts`: exactly that range's facts, the whole batch's counts known without building), and the live
    reference ReaRev behind `shard.shard_model` returns the unsharded pred_dist / pred / loss on both gloo ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_local_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err in res:
        assert ok, "rank %d: rank-local batch differs (max err %g)" % (rank, err)


def _train_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import copy
        import test_dropin_with_reference as td
        from gnnrag_amd import shard
        args, dataset, model = td.build_reference_setup()
        train = dataset["train"]
        train.reset_batches(is_sequential=True)
        np.random.seed(3)
        batch = train.get_batch(0, 5, fact_dropout=0.0)
        model.train()
        for m in model.modules():                       # dropout off: the two runs must see the same network
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        ref = copy.deepcopy(model)
        loss_ref = ref(batch, training=True)[0]
        loss_ref.backward()
        loss, _ = shard.sharded_training_step(model, batch)
        # (some gradients are mathematically zero - score_func.bias: softmax is shift invariant - so errors are
        # measured against the largest gradient of the model, not per tensor)
        scale = max(float(pr.grad.abs().max()) for pr in ref.parameters() if pr.grad is not None)
        worst = 0.0
        for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
            if pr.grad is None:
                continue
            worst = max(worst, float((p.grad - pr.grad).abs().max()) / scale)
        ok = worst <= 1e-4 and abs(float(loss) - float(loss_ref)) <= 1e-5 * max(1.0, abs(float(loss_ref)))
        # gradient accumulation: a second step without zeroing adds ONE more whole-batch gradient (what .grad held
        # on entry is not summed over the ranks again)
        shard.sharded_training_step(model, batch)
        for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
            if pr.grad is None:
                continue
            worst = max(worst, float((p.grad - 2 * pr.grad).abs().max()) / scale)
        ok = ok and worst <= 2e-4
        q.put((rank, bool(ok), worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(not os.path.isdir(REF), reason="live reference not available")
def test_sharded_training_step_world2():
    """Data-parallel training step of the live reference ReaRev (CPU) on 2 gloo ranks: per-rank loss on a
    fact-balanced question shard, ONE all-reduce of the flat gradient - every rank ends with the gradient (and loss)
    of the whole-batch step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err in res:
        assert ok, "rank %d: sharded gradient differs (max rel err %g)" % (rank, err)
