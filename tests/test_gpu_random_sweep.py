"""Seeded random shapes through the whole stack (structure, both forward paths, backward) against the float64
oracles: odd N / D / I combinations, questions without facts, duplicate facts, large and tiny relation
vocabularies, normalised and unnormalised weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5
RTOL = 5e-6       # unnormalised hub sums reach 10^2: fp32 rounding of a few hundred terms is relative


def _random_cfg(rng, k):
    from gnnrag_amd import synth
    D = int(rng.choice([8, 20, 48, 64, 100, 132, 200, 260]))
    N = int(rng.integers(3, 90))
    return synth.GraphConfig(
        name="rand%d" % k, B=int(rng.integers(1, 6)), N=N, E=int(rng.integers(0, 6 * N)), R=int(rng.choice([1, 3, 17, 300, 2500])),
        D=D, I=int(rng.integers(1, 6)), L=2, T=1, seed=int(rng.integers(1, 10 ** 6)),
        zipf_heads=bool(rng.integers(0, 2)), self_loop=bool(rng.integers(0, 4) > 0),
        normalized_gnn=bool(rng.integers(0, 2)), pos_emb=bool(rng.integers(0, 2)),
        n_real_min=0 if rng.integers(0, 3) == 0 else None,
        rel_per_question=int(rng.integers(2, 40)) if rng.integers(0, 2) else None)


@pytest.mark.parametrize("k", range(int(__import__("os").environ.get("GNNRAG_SWEEP_CASES", "48"))))
def test_random_shape(k):
    import gnnrag_amd  # noqa: F401
    import oracle.rearev_grad as og
    import oracle.rearev_np64 as onp
    from gnnrag_amd import ops, stack, synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1000 + k)
    cfg = _random_cfg(rng, k)
    batch = synth.make_batch(cfg)
    if k % 4 == 1 and batch.F > 4:                      # duplicate a few facts (the reference sums them twice too)
        et = list(batch.edge_tuple)
        dup = rng.integers(0, batch.F, size=3)
        for i in range(3):
            et[i] = np.concatenate([et[i], et[i][dup]])
        et[3] = np.concatenate([et[3], et[3][dup]])
        et[4] = np.arange(len(et[0]), dtype=np.int64)
        et[5], et[6] = synth.edge_weights(et[0], et[1], cfg.R1)
        batch.edge_tuple = tuple(et)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    want = onp.run_stack(batch, feats, params, use_type_layer=True, norm_rel=cfg.normalized_gnn)
    for path in (1, 2):
        got = stack.run_stack(batch, feats, params, dev, use_type_layer=True, norm_rel=cfg.normalized_gnn, path=path)
        np.testing.assert_allclose(got["h0"], want["h0"], rtol=RTOL, atol=TOL, err_msg="%s h0" % cfg)
        for c in range(cfg.T * cfg.L):
            np.testing.assert_allclose(got["h"][c], want["h"][c], rtol=RTOL, atol=TOL, err_msg="%s h %d path %d" % (cfg, c, path))
            np.testing.assert_allclose(got["dist"][c], want["dist"][c], rtol=RTOL, atol=TOL,
                                       err_msg="%s dist %d path %d" % (cfg, c, path))
    # backward of the aggregation, both forms
    B, N, D, I = cfg.B, cfg.N, cfg.D, cfg.I
    et = batch.edge_tuple
    T_f = rng.standard_normal((cfg.R1, D)).astype(np.float32)
    T_i = rng.standard_normal((cfg.R1, D)).astype(np.float32)
    ins = rng.standard_normal((B, I, D)).astype(np.float32)
    g_agg = rng.standard_normal((B * N, 2 * I * D)).astype(np.float32)
    prior = rng.random((B, N)).astype(np.float32)
    plan = ops.CsrPlan(et[0], et[1], et[2], B, N, cfg.R1, dev)
    weight = None
    if cfg.normalized_gnn:
        plan.attach_w_gnn(et[5])
        weight = et[5]
    _, gd_w, gi_w, gtf_w, gti_w = og.aggregate_grads(et, B, N, prior, ins, T_f, T_i, g_agg, weight)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for gather in (True, False):
        gd, gi, gtf, gti = ops.aggregate_backward(plan, t(prior), t(ins), t(T_f), t(T_i), t(g_agg), gather=gather)
        for name, got_t, want_a in (("g_dist", gd, gd_w), ("g_ins", gi, gi_w), ("g_T_fwd", gtf, gtf_w), ("g_T_inv", gti, gti_w)):
            np.testing.assert_allclose(got_t.cpu().numpy(), want_a, rtol=0, atol=TOL * max(np.abs(want_a).max(), 1e-6),
                                       err_msg="%s %s gather=%s" % (cfg, name, gather))


@pytest.mark.parametrize("k", range(8))
def test_random_shape_bf16x3_math(k):
    """The optional exact-split bf16x3 math mode of the dense projections on random shapes (forward, both paths)."""
    import gnnrag_amd  # noqa: F401
    import oracle.rearev_np64 as onp
    from gnnrag_amd import ops, stack, synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5000 + k)
    cfg = _random_cfg(rng, 100 + k)
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    want = onp.run_stack(batch, feats, params)
    old = ops.set_dense_math(ops.MATH_BF16X3)
    try:
        for path in (1, 2):
            got = stack.run_stack(batch, feats, params, dev, path=path)
            for c in range(cfg.T * cfg.L):
                np.testing.assert_allclose(got["h"][c], want["h"][c], rtol=RTOL, atol=TOL, err_msg="%s h %d" % (cfg, c))
                np.testing.assert_allclose(got["dist"][c], want["dist"][c], rtol=RTOL, atol=TOL, err_msg="%s dist %d" % (cfg, c))
    finally:
        ops.set_dense_math(old)


@pytest.mark.parametrize("k", range(int(__import__("os").environ.get("GNNRAG_SWEEP_LARGE", "6"))))
def test_random_large_shape(k):
    """Random shapes LARGE enough for the W-resident bf16x3 kernels (>= 8192 node rows, >= 1024 compact relation rows,
    hidden size 200 or 208) - ragged node counts, 1-3 instructions, normalised weights, pos_emb, Freebase-like
    vocabularies - through both kernel paths in the default (mixed) math mode against the float64 oracle."""
    import gnnrag_amd  # noqa: F401
    import oracle.rearev_np64 as onp
    from gnnrag_amd import ops, stack, synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7000 + k)
    N = int(rng.integers(900, 2300))
    B = int(rng.integers(max(4, 8192 // N + 1), 14))
    used = int(rng.integers(150, 500)) if rng.integers(0, 2) else None
    cfg = synth.GraphConfig(name="large%d" % k, B=B, N=N, E=int(rng.integers(3 * N, 7 * N)),
                            R=int(rng.choice([300, 650, 3000])) if used is None else 3000, D=int(rng.choice([200, 208])),
                            I=int(rng.integers(1, 4)), L=2, T=1, seed=int(rng.integers(1, 10 ** 6)),
                            zipf_heads=bool(rng.integers(0, 2)), normalized_gnn=bool(rng.integers(0, 2)),
                            pos_emb=bool(rng.integers(0, 2)), n_real_min=N // 2, rel_per_question=used)
    batch = synth.make_batch(cfg)
    et = batch.edge_tuple
    plan = ops.CsrPlan(et[0], et[1], et[2], cfg.B, cfg.N, cfg.R1, dev)
    assert cfg.B * cfg.N >= 8192 and plan.rel_total >= 1024, cfg
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    want = onp.run_stack(batch, feats, params, use_type_layer=False)
    for path in (2, 1):
        got = stack.run_stack(batch, feats, params, dev, use_type_layer=False, path=path)
        for c in range(cfg.T * cfg.L):
            scale = max(1.0, float(np.abs(want["h"][c]).max()))
            np.testing.assert_allclose(got["h"][c], want["h"][c], rtol=0, atol=TOL * scale,
                                       err_msg="%s h %d path %d" % (cfg, c, path))
            np.testing.assert_allclose(got["dist"][c], want["dist"][c], rtol=0, atol=TOL,
                                       err_msg="%s dist %d path %d" % (cfg, c, path))
