"""GPU tests of the backward kernels: kernel level against the float64
autograd restatement (oracle/rearev_grad.py), module level against gradients recorded from the live
reference (tests/golden/grad_*.npz).  Tolerances: 2e-5 of the largest entry per tensor at kernel level
(fp32 sums in atomic order), 3e-4 at module level (fp32 chains through up to 6 layers on both sides)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
TOL_KERNEL = 2e-5
TOL_MODULE = 3e-4


@pytest.fixture(scope="module")
def dev():
    import gnnrag_amd  # noqa: F401
    return torch.device("cuda", 0)


def _dev(dev, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) for a in arrs]


def _close(got, want, tol, msg, floor=1e-6):
    np.testing.assert_allclose(got, want, rtol=0, atol=tol * max(np.abs(want).max(), floor), err_msg=msg)


CASES = {
    "hub": dict(B=3, N=600, E=4000, R=20, D=200, I=2, L=1, seed=3),            # wave-per-node rows, heavy chunks
    "huge": dict(B=2, N=500, E=14000, R=20, D=200, I=2, L=1, seed=4),          # a row with > 4096 facts
    "odd": dict(B=2, N=33, E=150, R=5, D=30, I=3, L=1, seed=5, normalized_gnn=True),   # D % 4 != 0, three instructions
    "wide": dict(B=2, N=40, E=160, R=4, D=300, I=1, L=1, seed=6),              # D > 256: several columns per lane
}


def _cfg(name):
    from gnnrag_amd import synth
    return synth.CONFIGS[name] if name in synth.CONFIGS else synth.GraphConfig(name=name, **CASES[name])


@pytest.mark.parametrize("gather", [True, False], ids=["gather", "lds"])
@pytest.mark.parametrize("name", ["tiny", "tiny50", "tinyfb", "hub", "huge", "odd", "wide"])
def test_aggregate_backward_vs_f64_autograd(dev, name, gather):
    import oracle.rearev_grad as og
    from gnnrag_amd import ops, synth
    cfg = _cfg(name)
    batch = synth.make_batch(cfg)
    B, N, D, I = cfg.B, cfg.N, cfg.D, cfg.I
    et = batch.edge_tuple
    rng = np.random.default_rng(17)
    T_f = (0.5 * rng.standard_normal((cfg.R1, D))).astype(np.float32)
    T_i = (0.5 * rng.standard_normal((cfg.R1, D))).astype(np.float32)
    ins = (0.5 * rng.standard_normal((B, I, D))).astype(np.float32)
    g_agg = rng.standard_normal((B * N, 2 * I * D)).astype(np.float32)
    dense = rng.random((B, N)).astype(np.float32)
    dense /= dense.sum(1, keepdims=True)
    plan = ops.CsrPlan(et[0], et[1], et[2], B, N, cfg.R1, dev)
    weight = None
    if cfg.normalized_gnn:
        plan.attach_w_gnn(et[5])
        weight = et[5]
    for prior in (dense, batch.seed_dist.astype(np.float32)):
        agg_w, gd_w, gi_w, gtf_w, gti_w = og.aggregate_grads(et, B, N, prior, ins, T_f, T_i, g_agg, weight)
        d_prior, d_ins, d_tf, d_ti, d_g = _dev(dev, prior, ins, T_f, T_i, g_agg)
        agg = ops.aggregate(plan, d_prior, d_ins, d_tf, d_ti)
        _close(agg.cpu().numpy(), agg_w, TOL_KERNEL, "agg")
        gd, gi, gtf, gti = ops.aggregate_backward(plan, d_prior, d_ins, d_tf, d_ti, d_g, gather=gather)
        if gather and D % 4 == 0:
            # no atomics on this path: a second run gives the same bits for every gradient
            gd2, gi2, gtf2, gti2 = ops.aggregate_backward(plan, d_prior, d_ins, d_tf, d_ti, d_g, gather=True)
            assert torch.equal(gd, gd2) and torch.equal(gi, gi2) and torch.equal(gtf, gtf2) and torch.equal(gti, gti2)
        _close(gd.cpu().numpy(), gd_w, TOL_KERNEL, "g_dist")
        _close(gi.cpu().numpy(), gi_w, TOL_KERNEL, "g_ins")
        _close(gtf.cpu().numpy(), gtf_w, TOL_KERNEL, "g_T_fwd")
        _close(gti.cpu().numpy(), gti_w, TOL_KERNEL, "g_T_inv")
        # rows of relations no fact uses get exactly zero
        unused = np.setdiff1d(np.arange(cfg.R1), np.asarray(et[1]))
        assert not gtf.cpu().numpy()[unused].any() and not gti.cpu().numpy()[unused].any()


@pytest.mark.parametrize("gather", [True, False], ids=["gather", "lds"])
@pytest.mark.parametrize("name", ["tiny50", "tinyfb", "hub", "odd"])
@pytest.mark.parametrize("norm_rel", [False, True])
def test_typelayer_backward_vs_f64_autograd(dev, name, norm_rel, gather):
    import oracle.rearev_grad as og
    from gnnrag_amd import ops, synth
    cfg = _cfg(name)
    batch = synth.make_batch(cfg)
    et = batch.edge_tuple
    B, N, D = cfg.B, cfg.N, cfg.D
    rng = np.random.default_rng(23)
    T = rng.standard_normal((cfg.R1, D)).astype(np.float32)
    g_pre = rng.standard_normal((B * N, D)).astype(np.float32)
    plan = ops.CsrPlan(et[0], et[1], et[2], B, N, cfg.R1, dev)
    if norm_rel:
        plan.attach_w_rel(et[6])
    want = og.typelayer_grad(et, B, N, T, g_pre, et[6] if norm_rel else None)
    (d_g,) = _dev(dev, g_pre)
    got = ops.typelayer_backward(plan, d_g, norm_rel, gather=gather)
    _close(got.cpu().numpy(), want, TOL_KERNEL, "g_T")
    if gather and D % 4 == 0:
        assert torch.equal(got, ops.typelayer_backward(plan, d_g, norm_rel, gather=True))   # no atomics: same bits


def test_empty_batch_backward(dev):
    from gnnrag_amd import ops
    z = np.zeros(0, np.int64)
    plan = ops.CsrPlan(z, z, z, 2, 8, 3, dev)
    D, I = 16, 2
    dist = torch.full((2, 8), 0.125, device=dev)
    ins = torch.randn(2, I, D, device=dev)
    T = torch.randn(3, D, device=dev)
    g = torch.randn(16, 2 * I * D, device=dev)
    outs = ops.aggregate_backward(plan, dist, ins, T, T, g)
    assert all(not o.cpu().numpy().any() for o in outs)
    assert not ops.typelayer_backward(plan, torch.randn(16, D, device=dev), False).cpu().numpy().any()


@pytest.mark.parametrize("name", ["tiny", "tinyfb", "hub", "huge", "wide", "norm"])
def test_fused_walk_backward_vs_f64(dev, name):
    """gnnrag_aggregate_fused_backward (training on the fused form): g_dist and g_P against float64 sums over the caller's
    fact tuple - light rows, rows of more than 256 and more than 4096 facts, a Freebase-sized vocabulary with few
    relations per question, D > 256, normalized_gnn weights (applied twice, as the forward does); the forward value
    against the same sums; two runs bit-identical (gather kernels, fixed order)."""
    from gnnrag_amd import ops, synth
    cfg = (synth.GraphConfig(name="norm", B=2, N=60, E=300, R=9, D=64, I=2, L=1, seed=8, normalized_gnn=True)
           if name == "norm" else _cfg(name))
    batch = synth.make_batch(cfg)
    B, N, D = cfg.B, cfg.N, cfg.D
    et = batch.edge_tuple
    h, r, t = (np.asarray(et[k]).astype(np.int64) for k in range(3))
    plan = ops.CsrPlan(h, r, t, B, N, cfg.R1, dev)
    w = np.ones(len(h))
    if cfg.normalized_gnn:
        plan.attach_w_gnn(et[5])
        w = np.asarray(et[5], dtype=np.float64) ** 2                  # reasongnn.py:80,84: the weight enters twice
    rows = plan.to_host()["rel_rows"].astype(np.int64)
    key = rows[:, 0] * (cfg.R1 + 1) + rows[:, 1]
    row_of = np.searchsorted(key, (h // N) * (cfg.R1 + 1) + r)
    assert (key[row_of] == (h // N) * (cfg.R1 + 1) + r).all()
    rng = np.random.default_rng(23)
    dist = rng.random((B, N)).astype(np.float32)
    dist[:, ::5] = 0.0
    P = (0.3 * rng.standard_normal((2, plan.rel_total, D))).astype(np.float32)
    g = rng.standard_normal((B * N, D)).astype(np.float32)
    d64, P64, g64 = dist.reshape(-1).astype(np.float64), P.astype(np.float64), g.astype(np.float64)
    want_nbr = np.zeros((B * N, D))
    np.add.at(want_nbr, t, (w * d64[h])[:, None] * P64[0][row_of])
    np.add.at(want_nbr, h, (w * d64[t])[:, None] * P64[1][row_of])
    want_gd = np.zeros(B * N)
    np.add.at(want_gd, h, w * np.einsum("fd,fd->f", g64[t], P64[0][row_of]))
    np.add.at(want_gd, t, w * np.einsum("fd,fd->f", g64[h], P64[1][row_of]))
    want_gP = np.zeros_like(P64)
    np.add.at(want_gP[0], row_of, (w * d64[h])[:, None] * g64[t])
    np.add.at(want_gP[1], row_of, (w * d64[t])[:, None] * g64[h])
    d_dist, d_P, d_g = _dev(dev, dist, P, g)
    _close(ops.aggregate_fused(plan, d_dist, d_P).cpu().numpy(), want_nbr, TOL_KERNEL, "nbr")
    got_gd, got_gP = ops.aggregate_fused_backward(plan, d_dist, d_P, d_g)
    _close(got_gd.cpu().numpy(), want_gd, TOL_KERNEL, "g_dist")
    _close(got_gP.cpu().numpy(), want_gP, TOL_KERNEL, "g_P")
    again = ops.aggregate_fused_backward(plan, d_dist, d_P, d_g)
    assert torch.equal(got_gd, again[0]) and torch.equal(got_gP, again[1])


def test_dense_relation_tables_match_the_kernel(dev):
    """autograd.relation_tables_dense (the differentiable torch expression training uses) computes what
    gnnrag_relation_tables computes (exact-fp32 mode), rows in the structure's compact order."""
    from gnnrag_amd import ops, synth
    from gnnrag_amd.autograd import relation_tables_dense
    cfg = _cfg("tinyfb")
    batch = synth.make_batch(cfg)
    et = batch.edge_tuple
    plan = ops.CsrPlan(et[0], et[1], et[2], cfg.B, cfg.N, cfg.R1, dev)
    g = torch.Generator().manual_seed(3)
    r = lambda *sh: (0.4 * torch.randn(*sh, generator=g)).to(dev)
    Tf, Ti, ins, W = r(cfg.R1, cfg.D), r(cfg.R1, cfg.D), r(cfg.B, cfg.I, cfg.D), r(cfg.D, (2 * cfg.I + 1) * cfg.D)
    old = ops.set_dense_math(ops.MATH_FP32)
    try:
        want = ops.relation_tables(plan, Tf, Ti, ins, W)
    finally:
        ops.set_dense_math(old)
    got = relation_tables_dense(plan, Tf, Ti, ins, W)
    _close(got.cpu().numpy(), want.cpu().numpy(), TOL_KERNEL, "P")


@pytest.mark.parametrize("form", ["fused", "unfused"])
@pytest.mark.parametrize("name", ["layer_d200.npz", "layer_d50.npz"])
def test_module_gradients_match_reference_fixture(dev, name, form):
    """Our ReasonGNNLayer with autograd enabled: same loss as tests/golden/make_golden_grad.py, gradients
    of every parameter and input against what autograd gave through the live reference module - on the fused training
    form and on the unfused one."""
    from gnnrag_amd import stack
    cfg, batch, feats, params, ref = load_golden(name)
    z = np.load(os.path.join(GOLDEN, "grad_" + name))
    layer = stack.build_layer(cfg, batch, params, dev).train()
    # "fused": relation tables + fused walk with its own backward (the default without active dropout, hidden sizes that
    # are a multiple of 4 - layer_d50 falls back to the unfused form by itself); "unfused": AggregateFn as in round 2
    layer.train_fused = form == "fused"
    inp = {k: torch.tensor(feats[k], device=dev, requires_grad=True)
           for k in ("h0", "rel_features", "rel_features_inv", "ins")}
    layer.init_reason(local_entity=torch.from_numpy(batch.local_entity).to(dev), kb_adj_mat=batch.edge_tuple,
                      local_entity_emb=inp["h0"], rel_features=inp["rel_features"],
                      rel_features_inv=inp["rel_features_inv"],
                      query_entities=torch.from_numpy(batch.query_entities).float().to(dev))
    seed = torch.from_numpy(batch.seed_dist).float().to(dev)
    Gd = torch.from_numpy(z["cot.Gd"]).to(dev)
    Gh = torch.from_numpy(z["cot.Gh"]).to(dev)
    loss, c = 0.0, 0
    for t in range(cfg.T):
        dist = seed
        for j in range(cfg.L):
            dist, h = layer(dist, inp["ins"][t], step=j)
            assert np.abs(dist.detach().cpu().numpy() - ref["dist"][c]).max() <= 1e-4      # forward parity too
            loss = loss + (dist * Gd[c]).sum()
            c += 1
    loss = loss + (h * Gh).sum()
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) <= 1e-4 * max(1.0, abs(float(z["loss"])))
    got = {k: v.grad for k, v in inp.items()}
    got.update({k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
    names = [k[5:] for k in z.files if k.startswith("grad.")]
    for k in names:
        assert k in got and got[k] is not None, "no gradient for " + k
        # floor: score_func.bias has a mathematically zero gradient (softmax is shift invariant)
        _close(got[k].cpu().numpy(), z["grad." + k], TOL_MODULE, k, floor=1e-3)


@pytest.mark.parametrize("norm_rel", [False, True])
def test_type_layer_gradients_match_reference_fixture(dev, norm_rel):
    from gnnrag_amd import stack, synth
    z = np.load(os.path.join(GOLDEN, "typelayer.npz"))
    zg = np.load(os.path.join(GOLDEN, "grad_typelayer.npz"))
    B, N, D = int(z["B"]), int(z["N"]), int(z["D"])
    F = len(z["heads"])
    et = (z["heads"], z["rels"], z["tails"], z["batch_ids"], np.arange(F, dtype=np.int64),
          z["weight_list"].tolist(), z["weight_rel_list"].tolist())
    cfg = synth.GraphConfig(name="tl", B=B, N=N, D=D, R=int(z["R1"]) - 2)
    params = {"type_layer.kb_self_linear.weight": z["param.type_layer.kb_self_linear.weight"],
              "type_layer.kb_self_linear.bias": z["param.type_layer.kb_self_linear.bias"]}
    tl = stack.build_type_layer(cfg, params, dev, norm_rel).train()
    rf = torch.tensor(z["feat.rel_features"], device=dev, requires_grad=True)
    h0 = tl(local_entity=torch.from_numpy(z["local_entity"]).to(dev), edge_list=et, rel_features=rf)
    assert np.abs(h0.detach().cpu().numpy() - z["ref.h0_norm%d" % int(norm_rel)]).max() <= 1e-4
    (h0 * torch.from_numpy(zg["cot.G"]).to(dev)).sum().backward()
    tag = "grad.norm%d." % int(norm_rel)
    _close(rf.grad.cpu().numpy(), zg[tag + "rel_features"], TOL_MODULE, "rel_features")
    _close(tl.kb_self_linear.weight.grad.cpu().numpy(), zg[tag + "kb_self_linear.weight"], TOL_MODULE, "weight")
    _close(tl.kb_self_linear.bias.grad.cpu().numpy(), zg[tag + "kb_self_linear.bias"], TOL_MODULE, "bias")


def test_training_mode_with_dropout_runs_and_is_stochastic(dev):
    """linear_dropout > 0 in training mode (the reference default, parsing.py:37) goes through the autograd
    form: finite outputs, gradients for every used parameter, two passes differ."""
    from gnnrag_amd import stack
    cfg, batch, feats, params, _ = load_golden("layer_d200.npz")
    layer = stack.build_layer(cfg, batch, params, dev).train()
    layer.linear_dropout = 0.2
    layer.linear_drop.p = 0.2
    devin = stack.DeviceInputs(batch, feats, dev)
    outs = []
    for rep in range(2):
        layer.zero_grad()
        layer.init_reason(local_entity=devin.local_entity, kb_adj_mat=batch.edge_tuple, local_entity_emb=devin.h0,
                          rel_features=devin.rel_features, rel_features_inv=devin.rel_features_inv,
                          query_entities=devin.query_entities)
        dist = devin.seed_dist
        for j in range(cfg.L):
            dist, h = layer(dist, devin.ins[0], step=j)
        (dist * dist).sum().backward()
        assert torch.isfinite(dist).all()
        for k, p in layer.named_parameters():
            if k.startswith(("rel_linear", "e2e_linear", "score_func")):
                assert p.grad is not None and torch.isfinite(p.grad).all(), k
        outs.append(dist.detach().cpu().numpy())
    assert np.abs(outs[0] - outs[1]).max() > 0


@pytest.mark.parametrize("M,N1,N2", [(1000, 200, 200), (128000, 200, 200), (4097, 200, 1000), (37, 8, 12), (1, 4, 4),
                                      (5000, 56, 56), (70000, 200, 400)])
def test_gemm_tn_weight_gradient_vs_fp64(dev, M, N1, N2):
    """gnnrag_gemm_tn: dW = dY^T X (the nn.Linear weight gradient, train_model.py:209-233) against float64, incl. row
    counts that are not a multiple of the chunk / k-step sizes and output sizes that are not a multiple of 64; the sum
    order is fixed (two runs agree bit for bit)."""
    from gnnrag_amd import ops
    rng = np.random.default_rng(M + N1)
    A = rng.standard_normal((M, N1)).astype(np.float32)
    B = rng.standard_normal((M, N2)).astype(np.float32)
    dA, dB = _dev(dev, A, B)
    got = ops.gemm_tn(dA, dB)
    want = A.astype(np.float64).T @ B.astype(np.float64)
    _close(got.cpu().numpy(), want, 1e-5, "gemm_tn")
    assert torch.equal(got, ops.gemm_tn(dA, dB))


@pytest.mark.parametrize("M,K,Nout,relu,bias", [(3000, 200, 200, True, True), (513, 1000, 200, False, True),
                                                (602, 200, 200, False, False), (9000, 200, 200, True, True)])
def test_linear_autograd_function_matches_torch(dev, M, K, Nout, relu, bias):
    """autograd.LinearFn (forward and dx on gnnrag_linear, dW on gnnrag_gemm_tn) against torch's own nn.Linear autograd
    in float64."""
    from gnnrag_amd.autograd import linear
    rng = np.random.default_rng(K + M)
    x = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((Nout, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(Nout).astype(np.float32)
    gy = rng.standard_normal((M, Nout)).astype(np.float32)
    tx, tW, tb = (torch.from_numpy(v).double().requires_grad_(True) for v in (x, W, b))
    ty = torch.nn.functional.linear(tx, tW, tb if bias else None)
    if relu:
        ty = torch.relu(ty)
    ty.backward(torch.from_numpy(gy).double())
    dx, dW, db, dgy = _dev(dev, x, W, b, gy)
    dx.requires_grad_(True); dW.requires_grad_(True); db.requires_grad_(True)
    y = linear(dx, dW, db if bias else None, relu)
    y.backward(dgy)
    _close(y.detach().cpu().numpy(), ty.detach().numpy(), TOL_KERNEL, "y")
    _close(dx.grad.cpu().numpy(), tx.grad.numpy(), TOL_KERNEL, "dx")
    _close(dW.grad.cpu().numpy(), tW.grad.numpy(), TOL_KERNEL, "dW")
    if bias:
        _close(db.grad.cpu().numpy(), tb.grad.numpy(), TOL_KERNEL, "db")
