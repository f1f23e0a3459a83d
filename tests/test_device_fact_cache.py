"""The device-resident per-question fact cache (reference: gnn/dataset_load.py:473-527,
gnn/modules/kg_reasoning/base_gnn.py:19-51).  Needs no reference checkout: the loader is a stub holding the attributes
the caches read, so the ``-m gpu`` test below runs on the GPU box (where /root/reference does not exist)."""
import numpy as np
import pytest


class _StubLoader:
    """The attributes of the reference's BasicDataLoader that the fact caches read (dataset_load.py:473-527)."""

    def __init__(self, rng, n_q=9, N=40, num_rel=13):
        self.max_local_entity, self.num_kb_relation = N, num_rel + 1
        self.data_eff, self.use_self_loop = False, True
        self.kb_adj_mats, self.global2local_entity_maps = {}, {}
        for q in range(n_q):
            n = int(rng.integers(3, N + 1))
            e = int(rng.integers(0, 4 * n))
            self.kb_adj_mats[q] = (rng.integers(0, n, e), rng.integers(0, num_rel, e), rng.integers(0, n, e))
            self.global2local_entity_maps[q] = {i: i for i in range(n)}


def test_device_cache_tuple_matches_host_cache_on_cpu_tensors():
    """BatchFacts (the lazily filled tuple of the device-resident cache) holds what FactCache.batch returns - checked
    with the cache's blocks kept as CPU tensors, which exercises the same code without a GPU."""
    import torch
    from gnnrag_amd.data.fact_mat import DeviceFactCache, FactCache
    ld = _StubLoader(np.random.default_rng(4))
    host, devc = FactCache(ld), DeviceFactCache(ld, torch.device("cpu"))
    for ids in ([0, 3, 4], [8, 1], [], [2]):
        a, b = host.batch(ids), devc.batch(ids)
        assert len(b) == 7 and tuple(b.hrt_device.shape) == (3, len(a[0]))
        for k in range(3):
            np.testing.assert_array_equal(b[k].numpy(), a[k])
        for k in range(3, 7):
            np.testing.assert_array_equal(np.asarray(b[k]), np.asarray(a[k]))
        assert len(b[4]) == len(a[4]) and len(list(b)) == 7 and len(b[:3]) == 3


def test_device_resident_tuple_shards_like_the_host_tuple():
    """shard.shard_edge_tuple on a BatchFacts (ids on the device) gives the questions' facts re-based exactly as it
    does for the host tuple (advisor finding, round 2: it used to call np.asarray on CUDA tensors)."""
    import torch
    from gnnrag_amd import shard
    from gnnrag_amd.data.fact_mat import DeviceFactCache, FactCache
    ld = _StubLoader(np.random.default_rng(5))
    host, devc = FactCache(ld), DeviceFactCache(ld, torch.device("cpu"))
    ids = [0, 3, 4, 8, 1]
    a, b = host.batch(ids), devc.batch(ids)
    for lo, hi in ((0, 5), (0, 2), (2, 5), (3, 3), (4, 5)):
        sa, sb = shard.shard_edge_tuple(a, ld.max_local_entity, lo, hi), shard.shard_edge_tuple(b, ld.max_local_entity, lo, hi)
        for k in range(3):
            np.testing.assert_array_equal(sb[k].numpy(), sa[k])
        for k in range(3, 7):
            np.testing.assert_array_equal(np.asarray(sb[k]), np.asarray(sa[k]))
    np.testing.assert_array_equal(shard.facts_per_question(b, len(ids)), shard.facts_per_question(a, len(ids)))


@pytest.mark.gpu
def test_structure_from_the_device_resident_cache_is_bit_identical():
    """f-1 (rest): per-question id blocks cached on the GPU, batch = device-side concatenation with node offsets;
    the structure built from it equals the one built from the host tuple bit for bit, and the batch is rebuilt from the
    cache without touching the questions' arrays again."""
    import torch
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import ops
    from gnnrag_amd.data.fact_mat import DeviceFactCache, FactCache
    dev = torch.device("cuda", 0)
    ld = _StubLoader(np.random.default_rng(7), n_q=12, N=300, num_rel=25)
    host, devc = FactCache(ld), DeviceFactCache(ld, dev)
    for ids in ([0, 5, 7, 11], [3, 3, 1], [9]):
        a, b = host.batch(ids), devc.batch(ids)
        B, N, R1 = len(ids), ld.max_local_entity, ld.num_kb_relation + 1
        pa = ops.CsrPlan(a[0], a[1], a[2], B, N, R1, dev).to_host()
        pb = ops.CsrPlan(None, None, None, B, N, R1, dev, hrt_device=b.hrt_device).to_host()
        for k in pa:
            if k == "big":
                assert all(np.array_equal(x, y) for x, y in zip(pa[k], pb[k]))
            else:
                np.testing.assert_array_equal(pa[k], pb[k], err_msg=k)
    assert sorted(devc._dev) == [0, 1, 3, 5, 7, 9, 11]      # every question uploaded once


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["small", "hubs_and_heavy_rows"])
def test_batch_structure_as_concatenation_of_cached_question_structures(shape):
    """f-1 as SURVEY.md section 8 words it: per-question destination-sorted structures cached on the GPU at first use,
    a batch's structure = their concatenation with offsets (gnnrag_csr_concat: no upload, no sort, no wait for the
    stream).  Bit-identical to the structure built from the batch tuple - every array incl. the merged record stream,
    the compact relation rows, the heavy / big node lists - for repeated and empty questions, and layer outputs
    computed on either structure are bit-identical too."""
    import torch
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import ops
    from gnnrag_amd.data.fact_mat import DeviceStructureCache, FactCache
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    if shape == "small":
        ld = _StubLoader(rng, n_q=12, N=300, num_rel=25)
        ld.kb_adj_mats[4] = (np.zeros(0, np.int64),) * 3                    # a question without typed edges
        batches = ([0, 5, 7, 11], [3, 3, 1], [9], [4, 2, 4])
    else:
        ld = _StubLoader(rng, n_q=5, N=1500, num_rel=40)
        for q in range(5):                                                   # hubs: rows of > 256 and > 4096 facts
            n = len(ld.global2local_entity_maps[q])
            e = 9000
            ld.kb_adj_mats[q] = ((rng.zipf(1.5, e) % n).astype(np.int64), rng.integers(0, 40, e), rng.integers(0, n, e))
        batches = ([0, 1, 2, 3, 4], [4, 0])
    host, devc = FactCache(ld), DeviceStructureCache(ld, dev)
    N, R1 = ld.max_local_entity, ld.num_kb_relation + 1
    for ids in batches:
        a, b = host.batch(ids), devc.batch(ids)
        B = len(ids)
        pa = ops.CsrPlan(a[0], a[1], a[2], B, N, R1, dev)
        pb = ops.CsrPlan.concat(b.plans, N, R1, dev)
        assert (pa.rel_total, pa.rel_max, pa.F) == (pb.rel_total, pb.rel_max, pb.F)
        ha, hb = pa.to_host(), pb.to_host()
        for k in ha:
            if k == "big":
                assert all(np.array_equal(x, y) for x, y in zip(ha[k], hb[k]))
            else:
                np.testing.assert_array_equal(ha[k], hb[k], err_msg=k)
        np.testing.assert_array_equal(b.hrt_device.cpu().numpy(), np.stack([a[0], a[1], a[2]]))     # lazily built id block
        # one fused layer on both structures
        D, I = 200, 2
        g = torch.Generator().manual_seed(B)
        r = lambda *sh: (0.3 * torch.randn(*sh, generator=g)).to(dev)
        dist = torch.rand(B, N, generator=g).to(dev)
        T0, T1, ins, W = r(R1, D), r(R1, D), r(B, I, D), r(D, (2 * I + 1) * D)
        outs = []
        for pl in (pa, pb):
            P = ops.relation_tables(pl, T0, T1, ins, W)
            outs.append(ops.aggregate_fused(pl, dist, P))
        assert torch.equal(outs[0], outs[1])
    assert len(devc._plans) == len({i for ids in batches for i in ids})      # every question sorted once
    # a rank's shard of a structure-cache batch (shard.shard_edge_tuple -> BatchFacts.shard): the questions' structures
    # and, lazily, the re-based id block
    from gnnrag_amd import shard
    ids = batches[0]
    a, b = host.batch(ids), devc.batch(ids)
    lo, hi = 1, len(ids)
    sa, sb = shard.shard_edge_tuple(a, N, lo, hi), shard.shard_edge_tuple(b, N, lo, hi)
    pa = ops.CsrPlan(sa[0], sa[1], sa[2], hi - lo, N, R1, dev).to_host()
    pb = ops.CsrPlan.concat(sb.plans, N, R1, dev).to_host()
    for k in pa:
        if k != "big":
            np.testing.assert_array_equal(pa[k], pb[k], err_msg="shard " + k)
    np.testing.assert_array_equal(sb.hrt_device.cpu().numpy(), np.stack([sa[0], sa[1], sa[2]]))


@pytest.mark.gpu
def test_build_told_the_relation_counts_does_not_wait_and_is_the_same_structure():
    """`gnnrag_csr_build_counts` (ABI 15): a device fact cache knows every question's distinct-relation count, so the
    batch's build is told (rel_total, rel_max) and returns without waiting for its stream - bit-identical to the waiting
    build of the same tuple (`plan_for` takes that form for a `BatchFacts`), `status()` confirms tuple and counts on the
    device; wrong counts and an invalid tuple are reported by `status()`."""
    import torch
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import _lib, ops
    from gnnrag_amd.data.fact_mat import DeviceFactCache, FactCache
    from gnnrag_amd.modules.kg_reasoning.base_gnn import plan_for
    dev = torch.device("cuda", 0)
    ld = _StubLoader(np.random.default_rng(5), n_q=9, N=300, num_rel=40)
    host, devc = FactCache(ld), DeviceFactCache(ld, dev)
    N, R1 = ld.max_local_entity, ld.num_kb_relation + 1
    ids = [0, 3, 3, 8, 5]
    a, bf = host.batch(ids), devc.batch(ids)
    want_plan = ops.CsrPlan(a[0], a[1], a[2], len(ids), N, R1, dev)
    assert bf.rel_counts == (want_plan.rel_total, want_plan.rel_max)
    got_plan = plan_for(bf, len(ids), N, R1, dev)
    got_plan.status()
    want, got = want_plan.to_host(), got_plan.to_host()
    assert (got_plan.rel_total, got_plan.rel_max) == (want_plan.rel_total, want_plan.rel_max)
    for key in want:
        if key == "big":
            assert all(np.array_equal(x, y) for x, y in zip(want[key], got[key]))
        else:
            np.testing.assert_array_equal(want[key], got[key], err_msg=key)
    # counts that are not the device's
    bad = ops.CsrPlan(None, None, None, len(ids), N, R1, dev, hrt_device=bf.hrt_device,
                      rel_counts=(bf.rel_counts[0] + 1, bf.rel_counts[1]))
    with pytest.raises(_lib.GnnragError):
        bad.status()
    # an invalid tuple (a fact across two questions) passes the no-wait build and fails the deferred check
    hrt = bf.hrt_device.clone()
    hrt[2, 0] = N + 1
    bad = ops.CsrPlan(None, None, None, len(ids), N, R1, dev, hrt_device=hrt, rel_counts=bf.rel_counts)
    with pytest.raises(ValueError):
        bad.status()
