"""The frontier form of the first layer of a ReaRev iteration (GNNRAG_PATH_SEED_PRIOR, csrc/frontier.hip).

Reference semantics: rearev.py:208 resets ``curr_dist`` to the seed distribution at the top of every iteration, and
``fact_prior = head2fact . dist`` (reasongnn.py:80 / :106) is zero for every fact that does not start at a seed, so
``fact_val * fact_prior`` (:82) adds exact zeros.  The library therefore computes, for that layer, only the relation
table rows the seeds' facts use and the neighbour sums of the nodes they reach.  Checked here:

* the frontier itself (row gates, list sizes) against a numpy statement of "nodes reached by a fact whose source has
  dist != 0", for one seed, several seeds, hub seeds, a dense prior and more seeds than the kernel lists in LDS;
* the listed relation-table rows against the full table launch and the float64 definition; unlisted rows untouched;
* the listed neighbour sums against the full fused walk on the same prior - and the full walk's unlisted rows are
  exactly zero, which is what makes skipping them legal;
* whole layer stacks with and without the hint (module switch GNNRAG_SEED_PRIOR) agree to fp32 rounding, for shapes
  that take each of the three update kernels (bf16x3 W-resident, fp32 W-resident, k-tiled + memset fallback).
All other GPU parity tests run with the hint ON (the module's default), i.e. they pin the frontier path to the oracles.
"""
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


def _plan_of(batch, dev):
    from gnnrag_amd import ops
    cfg = batch.cfg
    et = batch.edge_tuple
    return ops.CsrPlan(et[0], et[1], et[2], cfg.B, cfg.N, cfg.R1, dev)


def _numpy_frontier(batch, dist):
    """Row gates: node n is on the frontier iff some fact with dst_d(f) = n has dist[src_d(f)] != 0 (d = 0: head ->
    tail, d = 1: tail -> head).  Relation rows: (question, relation) pairs of those facts."""
    heads, rels, tails = (np.asarray(x) for x in batch.edge_tuple[:3])
    nz = dist.reshape(-1) != 0
    BN = dist.size
    flag = np.zeros(BN, dtype=np.uint8)
    f0, f1 = nz[heads], nz[tails]
    flag[tails[f0]] = 1
    flag[heads[f1]] = 1
    N = batch.cfg.N
    pairs = set(zip((heads[f0] // N).tolist(), rels[f0].tolist())) | set(zip((heads[f1] // N).tolist(), rels[f1].tolist()))
    return flag, len(pairs)


def _cfg(**kw):
    from gnnrag_amd import synth
    base = dict(name="fr", B=6, N=500, E=2500, R=40, D=200, I=2, L=3, T=2, seed=99)
    base.update(kw)
    return synth.GraphConfig(**base)


@pytest.mark.parametrize("case", ["one_seed", "three_seeds", "hub_seed", "dense_prior", "more_seeds_than_the_lds_list",
                                  "weighted", "empty_question", "listed_hub", "listed_hub_weighted"])
def test_frontier_lists_tables_and_sums(dev, case):
    from gnnrag_amd import ops, synth
    import oracle.rearev_np64 as onp
    kw = {}
    if case == "more_seeds_than_the_lds_list":
        kw = dict(B=2, N=2300, E=6000)
    if case == "weighted":
        kw = dict(normalized_gnn=True)
    if case == "empty_question":
        kw = dict(n_real_min=0, B=7)
    if case.startswith("listed_hub"):          # a hub of > 4096 facts ON the frontier of low-degree seeds: the walk
        kw = dict(B=3, N=3000, E=40000, R=60, normalized_gnn=case.endswith("weighted"))   # resolves it from the seeds' rows
    cfg = _cfg(**kw)
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    rng = np.random.default_rng(5)
    dist = batch.seed_dist.astype(np.float32).copy()
    if case == "three_seeds":
        for b in range(cfg.B):
            dist[b] = 0
            dist[b, rng.choice(cfg.N, 3, replace=False)] = 1.0 / 3
    elif case == "hub_seed":
        dist[:] = 0
        dist[:, 1] = 1.0                        # node 1 is the Zipf hub: thousands of facts start there
    elif case.startswith("listed_hub"):
        heads, _, tails = (np.asarray(x) for x in batch.edge_tuple[:3])
        deg = np.bincount(heads, minlength=cfg.B * cfg.N) + np.bincount(tails, minlength=cfg.B * cfg.N)
        dist[:] = 0
        for b in range(cfg.B):
            hub = b * cfg.N + int(np.argmax(deg[b * cfg.N:(b + 1) * cfg.N]))
            assert deg[hub] > 4096
            nb = np.concatenate([tails[heads == hub], heads[tails == hub]])            # the hub's neighbours
            nb = nb[(nb != hub) & (deg[nb] < 64)]
            pick = np.unique(nb)[:2]                                                     # two low-degree seeds next to it
            assert len(pick) == 2
            dist[b, pick - b * cfg.N] = 0.5
    elif case == "dense_prior":
        dist = rng.random((cfg.B, cfg.N)).astype(np.float32)
        dist /= dist.sum(1, keepdims=True)
    elif case == "more_seeds_than_the_lds_list":
        dist = rng.random((cfg.B, cfg.N)).astype(np.float32)       # 2300 > 2048 non-zeros per question
        dist[1, 5:] = 0                          # ... in question 0; question 1 has five seeds
    plan = _plan_of(batch, dev)
    if cfg.normalized_gnn:
        plan.attach_w_gnn(batch.edge_tuple[5])
    d_dev = torch.from_numpy(dist).to(dev)
    fr = ops.Frontier(plan, d_dev)
    nrows, ntrows, flags = fr.read()
    want_flags, want_pairs = _numpy_frontier(batch, dist)
    if case == "more_seeds_than_the_lds_list":  # question 0 overflows the LDS seed list: flagged as a whole (a superset)
        N = cfg.N
        assert flags[:N].all() and (flags[N:] == want_flags[N:]).all()
    else:
        np.testing.assert_array_equal(flags, want_flags)
        assert ntrows == want_pairs
    assert nrows == int(flags.sum())

    # relation tables: listed rows equal the full launch's rows (exact fp32 vs the math mode's kernel: fp32 rounding),
    # unlisted rows keep the NaN fill
    T = ops.rel_transform(torch.from_numpy(feats["rel_features"]).to(dev), torch.from_numpy(feats["rel_features_inv"]).to(dev),
                          [(torch.from_numpy(params["rel_linear0.weight"]).to(dev), torch.from_numpy(params["rel_linear0.bias"]).to(dev), None, None)])
    ins = torch.from_numpy(feats["ins"][0]).to(dev)
    W = torch.from_numpy(params["e2e_linear0.weight"]).to(dev)
    P_full = ops.relation_tables(plan, T[0, 0], T[0, 1], ins, W, math=ops.MATH_FP32)
    P_fr = fr.relation_tables(T[0, 0], T[0, 1], ins, W)
    listed = ~torch.isnan(P_fr[0, :, 0])
    assert int(listed.sum()) == ntrows and torch.equal(listed, ~torch.isnan(P_fr[1, :, 0]))
    scale = max(1.0, float(P_full.abs().max()))
    assert float((P_fr[:, listed] - P_full[:, listed]).abs().max()) <= 2e-6 * scale

    # neighbour sums: the full walk's rows off the frontier are exactly zero; on it both walks agree
    nbr_full = ops.aggregate_fused(plan, d_dev, P_full)
    on = torch.from_numpy(flags.astype(bool)).to(dev)
    assert float(nbr_full[~on].abs().max()) == 0.0 if (~on).any() else True
    P_mix = torch.where(torch.isnan(P_fr), torch.zeros_like(P_fr), P_fr)
    nbr_fr = fr.aggregate(P_mix)
    assert float(nbr_fr[~on].abs().max()) == 0.0 if (~on).any() else True
    s = max(1.0, float(nbr_full.abs().max()))
    assert float((nbr_fr - nbr_full).abs().max()) <= 4e-6 * s
    # with the SAME table values both walks add a row's live facts in position order; the streaming walk cuts the merged
    # stream into equal fact ranges, so a row cut by a range boundary groups its sum differently: equal to rounding
    nbr_same = fr.aggregate(P_full)
    assert float((nbr_same - nbr_full).abs().max()) <= 2e-6 * s


@pytest.mark.parametrize("shape", ["b3", "wres56", "ktiled_small", "wide_slices", "pos_emb_norm"])
def test_layer_stack_with_and_without_the_seed_prior_hint(dev, shape, monkeypatch):
    """T x L layer calls through the drop-in module with GNNRAG_SEED_PRIOR on / off: same distributions, embeddings
    and argmax; and both against the float64 oracle."""
    import oracle.rearev_np64 as onp
    from gnnrag_amd import stack, synth
    if shape == "b3":            # >= 8192 rows, D = 200: bf16x3 W-resident update with row gates
        cfg = _cfg(B=5, N=2000, E=9000, R=300, T=2)
    elif shape == "wres56":      # D = 50 -> 56, >= 4096 rows: fp32 W-resident update with row gates
        cfg = _cfg(B=4, N=1500, E=5000, R=60, D=50, T=3, n_real_min=200)
    elif shape == "ktiled_small":  # < 4096 rows: k-tiled update, nbr zero-filled
        cfg = _cfg(B=3, N=300, E=1200, R=25, D=200, T=2)
    elif shape == "wide_slices":
        cfg = _cfg(B=9, N=700, E=3000, R=2000, D=200, T=2, rel_per_question=120)
    else:
        cfg = _cfg(B=4, N=400, E=1500, R=17, D=200, T=2, normalized_gnn=True, pos_emb=True, n_real_min=20)
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    want = onp.run_stack(batch, feats, params, use_type_layer=True)
    got = {}
    for hint in ("1", "0"):
        monkeypatch.setenv("GNNRAG_SEED_PRIOR", hint)
        got[hint] = stack.run_stack(batch, feats, params, dev, use_type_layer=True)
    for c in range(cfg.T * cfg.L):
        a, b = got["1"], got["0"]
        sc = max(1.0, np.abs(want["h"][c]).max())
        assert np.abs(a["h"][c] - b["h"][c]).max() <= 4e-6 * sc, (shape, c)
        assert np.abs(a["dist"][c] - b["dist"][c]).max() <= 4e-6, (shape, c)
        for g in (a, b):
            assert np.abs(g["h"][c] - want["h"][c]).max() <= 2e-5 * sc, (shape, c)
            assert np.abs(g["dist"][c] - want["dist"][c]).max() <= 2e-5, (shape, c)
            assert (g["dist"][c].argmax(1) == want["dist"][c].argmax(1)).all()
