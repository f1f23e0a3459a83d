"""The streaming form of the fused LDS walk (k_walk_stream, csrc/aggregate.hip; opt-in: GNNRAG_WALK_STREAM=1): the merged
record stream of a question cut into equal fact ranges, row boundaries found in the stream, cut rows completed from head /
tail partials in position order.  Measured slower than the set walk at C2 (DESIGN appendix), so it is not the default -
but it is a second, structurally different implementation of reasongnn.py:80-84 / :106-111 in the fused form, and these
tests pin it to the set walk and to the float64 definition: hub rows cut dozens of times, rows without facts, ragged
questions, few questions (node-range parts), 32-column slices, per-fact weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)


@pytest.mark.parametrize("case", ["c2_like", "hubs", "few_questions", "wide_slices", "ragged_weighted", "tiny"])
def test_streaming_walk_equals_set_walk_and_fp64(dev, case, monkeypatch):
    from gnnrag_amd import ops, synth
    kw = dict(name=case, B=6, N=2000, E=10000, R=600, D=200, I=2, L=1, T=1, seed=31)
    if case == "hubs":
        kw.update(B=3, N=1500, E=30000, R=500)                      # the Zipf hub holds > 10 000 facts: cut > 100 times
    elif case == "few_questions":
        kw.update(B=1, N=2000, E=8000, R=500)                       # node-range parts (one question fills an XCD)
    elif case == "wide_slices":
        kw.update(B=9, N=900, E=4000, R=3000, rel_per_question=150)  # <= 300 relations per question: 32-column slices
    elif case == "ragged_weighted":
        kw.update(B=7, N=700, E=3000, R=40, n_real_min=0, normalized_gnn=True)   # padded nodes have no facts: zero rows
    elif case == "tiny":
        kw.update(B=3, N=48, E=150, R=11)
    cfg = synth.GraphConfig(**kw)
    batch = synth.make_batch(cfg)
    et = batch.edge_tuple
    plan = ops.CsrPlan(et[0], et[1], et[2], cfg.B, cfg.N, cfg.R1, dev)
    if cfg.normalized_gnn:
        plan.attach_w_gnn(et[5])
    assert ops.aggregate_fused_variant(plan, cfg.D) == (ops.WALK_LDS_32 if case in ("wide_slices", "ragged_weighted", "tiny")
                                                         else ops.WALK_LDS_16)
    g = torch.Generator().manual_seed(5)
    dist = torch.rand(cfg.B, cfg.N, generator=g)
    dist = (dist / dist.sum(1, keepdim=True)).to(dev)
    P = (0.5 * torch.randn(2, plan.rel_total, cfg.D, generator=g)).to(dev)
    monkeypatch.delenv("GNNRAG_WALK_STREAM", raising=False)
    ref = ops.aggregate_fused(plan, dist, P)
    monkeypatch.setenv("GNNRAG_WALK_STREAM", "1")
    got = ops.aggregate_fused(plan, dist, P)
    got2 = ops.aggregate_fused(plan, dist, P)
    assert torch.equal(got, got2)                                    # one fixed summation order: bit-reproducible
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) <= 4e-6 * scale
    # float64 definition: nbr[n] = sum_d sum_{f: dst_d(f) = n} w_f dist[src_d(f)] P[d, row(b, rel_f)]
    h, r, t = (np.asarray(x) for x in et[:3])
    rows = plan.rel_rows()
    key = {(int(b), int(rel)): i for i, (b, rel) in enumerate(rows)}
    ridx = np.array([key[(int(hh) // cfg.N, int(rr))] for hh, rr in zip(h, r)])
    w = np.asarray(et[5], np.float64) ** 2 if cfg.normalized_gnn else np.ones(len(h))
    d64, P64 = dist.cpu().double().numpy().reshape(-1), P.cpu().double().numpy()
    want = np.zeros((cfg.B * cfg.N, cfg.D))
    np.add.at(want, t, (w * d64[h])[:, None] * P64[0, ridx])
    np.add.at(want, h, (w * d64[t])[:, None] * P64[1, ridx])
    assert np.abs(got.cpu().numpy() - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    deg = np.bincount(h, minlength=cfg.B * cfg.N) + np.bincount(t, minlength=cfg.B * cfg.N)
    assert (got.cpu().numpy()[deg == 0] == 0).all()                  # rows without facts are written as zeros
