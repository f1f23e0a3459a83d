"""Round-3 parity additions ("Next round" item 1, and the advisor's finding on k_gemm_wres):

* the self-block update (``gnnrag_update_score_fused``, reasongnn.py:161-168) at LARGE row counts for every hidden
  size class of the W-resident kernels - D = 32, 56, 64, 100, 128, 160, 200, 208 - in all three math modes against
  the float64 definition.  Round 2 only covered B*N >= 4096 at D = 200 / 208; D = 56 (the released checkpoints'
  entity_dim 50 zero-padded) overran the kernel's LDS block, D = 32 / 100 / 160 read k groups beyond K;
* BASELINE config C3 (B = 32, D = 50, T = 3, ragged n_real) through the drop-in module against both oracles;
* ONE full C2 batch (all 64 questions) against the torch-CPU oracle;
* the C1 shape with the kernel variant of the width the module REALLY runs (padded 56 -> LDS walk) and, explicitly,
  the unpadded scalar path (``pad_dim = False`` -> float2 gather walk).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_STATED = 1e-4
TOL_INTERNAL = 2e-5


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


def _check_stack(got, want, ncalls, tol=TOL_STATED, what=""):
    for c in range(ncalls):
        eh = np.abs(got["h"][c] - want["h"][c]).max() / max(1.0, np.abs(want["h"][c]).max())
        ed = np.abs(got["dist"][c] - want["dist"][c]).max()
        assert eh <= tol and ed <= tol, (what, c, eh, ed)
        assert (got["dist"][c].argmax(1) == want["dist"][c].argmax(1)).all(), (what, c)


@pytest.mark.parametrize("math", ["fp32", "mixed", "bf16x3"])
@pytest.mark.parametrize("D", [32, 56, 64, 100, 128, 160, 200, 208])
@pytest.mark.parametrize("M", [4096, 6000, 40003])
def test_self_block_update_large_m_every_hidden_size_vs_fp64(dev, M, D, math):
    """h' = relu(h W_self^T + b + nbr), score = w_s.h' + b_s + (1 - mask)(-1e11) (reasongnn.py:161-168) at row counts
    where the W-resident kernels are dispatched (M >= 4096), for hidden sizes in every (column tiles, k groups) class
    of k_gemm_wres incl. the ones between the compiled k-group counts and the padded released-checkpoint size 56."""
    from gnnrag_amd import ops
    I = 2
    g = torch.Generator(device="cpu").manual_seed(1000 * D + M % 997)
    r = lambda *shape: torch.randn(*shape, generator=g)
    h, nbr, W, b, ws, bs = r(M, D), r(M, D), r(D, (2 * I + 1) * D) / np.sqrt(D), r(D), r(D), r(1)
    mask = (torch.rand(M, generator=g) > 0.1).float()
    m = {"fp32": ops.MATH_FP32, "mixed": ops.MATH_MIXED, "bf16x3": ops.MATH_BF16X3}[math]
    h_out, score = ops.update_score_fused(h.to(dev), nbr.to(dev), W.to(dev), b.to(dev), ws.to(dev), bs.to(dev),
                                          mask.to(dev), I, math=m)
    want = np.maximum(h.double().numpy() @ W[:, :D].double().numpy().T + b.double().numpy() + nbr.double().numpy(), 0.0)
    want_s = want @ ws.double().numpy() + float(bs)
    got, got_s = h_out.cpu().numpy(), score.cpu().numpy()
    assert np.abs(got - want).max() <= TOL_INTERNAL * max(1.0, np.abs(want).max())
    live = mask.numpy() > 0
    assert np.abs(got_s[live] - want_s[live]).max() <= TOL_STATED * max(1.0, np.abs(want_s[live]).max())
    assert (got_s[~live] == np.float32(-1e11)).all()                  # fp32 add rounds to exactly -1e11


@pytest.mark.parametrize("M", [70001, 131072 + 17])
def test_self_block_update_very_large_ragged_rows_vs_fp64(dev, M):
    """k_update_b3 at row counts beyond 65 536 that are not a multiple of 16 or of the chunk count (round 4; also the shape
    size at which the kernel's 32-bit byte offsets are exercised near their upper rows): every row and every score against float64, masked scores
    exactly -1e11, two runs bit-identical."""
    from gnnrag_amd import ops
    D, I = 200, 2
    g = torch.Generator(device="cpu").manual_seed(M)
    r = lambda *shape: torch.randn(*shape, generator=g)
    h, nbr, W, b, ws, bs = r(M, D), r(M, D), r(D, (2 * I + 1) * D) / np.sqrt(D), r(D), r(D), r(1)
    mask = (torch.rand(M, generator=g) > 0.1).float()
    args = (h.to(dev), nbr.to(dev), W.to(dev), b.to(dev), ws.to(dev), bs.to(dev), mask.to(dev), I)
    h_out, score = ops.update_score_fused(*args, math=ops.MATH_MIXED)
    want = np.maximum(h.double().numpy() @ W[:, :D].double().numpy().T + b.double().numpy() + nbr.double().numpy(), 0.0)
    want_s = want @ ws.double().numpy() + float(bs)
    got, got_s = h_out.cpu().numpy(), score.cpu().numpy()
    assert np.abs(got - want).max() <= TOL_INTERNAL * max(1.0, np.abs(want).max())
    live = mask.numpy() > 0
    assert np.abs(got_s[live] - want_s[live]).max() <= TOL_STATED * max(1.0, np.abs(want_s[live]).max())
    assert (got_s[~live] == np.float32(-1e11)).all()
    h2, s2 = ops.update_score_fused(*args, math=ops.MATH_MIXED)
    assert torch.equal(h_out, h2) and torch.equal(score, s2)


@pytest.mark.parametrize("math", ["mixed", "fp32"])
def test_c3_shape_released_checkpoint_dims(dev, math):
    """BASELINE config C3 per GPU: 32 WebQSP-shaped ragged questions, released-checkpoint dims (entity_dim 50, 2
    instructions, 3 layers, 3 iterations) incl. TypeLayer, through the drop-in module: the hidden size is zero-padded
    to 56, so the LDS walk (32-column slices) and - B*N = 64 000 rows - the W-resident update kernel at Nout = 56 run,
    which is the case the advisor found broken in round 2 (LDS rows 56..61 of the ColMap order)."""
    import oracle.rearev_np64 as onp
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import ops, stack, synth
    cfg = synth.CONFIGS["C3"]
    old = ops.set_dense_math({"mixed": ops.MATH_MIXED, "fp32": ops.MATH_FP32}[math])
    try:
        batch = synth.make_batch(cfg)
        assert batch.n_real.min() < cfg.N // 2 < batch.n_real.max()   # ragged
        feats = synth.make_features(cfg)
        params = synth.make_layer_params(cfg)
        plan = _plan_of(batch, dev)
        assert ops.aggregate_fused_variant(plan, 56) in (ops.WALK_LDS_16, ops.WALK_LDS_32)
        want64 = onp.run_stack(batch, feats, params, use_type_layer=True)
        want = otorch.run_stack(batch, feats, params, use_type_layer=True)
        for path in (0, 2, 1):
            got = stack.run_stack(batch, feats, params, dev, use_type_layer=True, path=path)
            assert np.abs(got["h0"] - want64["h0"]).max() <= TOL_INTERNAL * max(1.0, np.abs(want64["h0"]).max())
            _check_stack(got, want64, cfg.T * cfg.L, tol=TOL_INTERNAL, what="C3 np64 path %d" % path)
            _check_stack(got, want, cfg.T * cfg.L, what="C3 torch path %d" % path)
    finally:
        ops.set_dense_math(old)


def test_c2_whole_batch_against_oracle(dev):
    """All 64 questions of the C2 batch bench.py times, default math mode, fused (= auto) path, against the torch-CPU
    oracle on the SAME whole batch (the round-2 test compared 4 of the 64 questions)."""
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import ops, stack, synth
    cfg = synth.CONFIGS["C2"]
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    plan = _plan_of(batch, dev)
    assert ops.aggregate_fused_variant(plan, cfg.D) == ops.WALK_LDS_16
    got = stack.run_stack(batch, feats, params, dev)
    want = otorch.run_stack(batch, feats, params)
    _check_stack(got, want, cfg.T * cfg.L, what="C2 whole batch")


def test_c1_kernel_variants_padded_and_unpadded(dev, monkeypatch):
    """C1 (B = 1, D = 50): the module pads the hidden size to 56 and runs the LDS walk; with GNNRAG_PAD_DIM=0 the
    library sees D = 50 and runs the scalar paths (table-row gather walk with float2 lanes, scalar-loader GEMMs).
    Both against both oracles, each asserting the variant of the width it really runs."""
    import oracle.rearev_np64 as onp
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import ops, stack, synth
    cfg = synth.CONFIGS["C1"]
    batch = synth.make_batch(cfg, seed=12)
    feats = synth.make_features(cfg, seed=12)
    params = synth.make_layer_params(cfg)
    plan = _plan_of(batch, dev)
    want64 = onp.run_stack(batch, feats, params, use_type_layer=True)
    want = otorch.run_stack(batch, feats, params, use_type_layer=True)
    for pad, width, variants in ((True, 56, (ops.WALK_LDS_16, ops.WALK_LDS_32)), (False, 50, (ops.WALK_L2_GATHER,))):
        monkeypatch.setenv("GNNRAG_PAD_DIM", "1" if pad else "0")
        assert ops.aggregate_fused_variant(plan, width) in variants
        for path in (0, 1, 2):
            got = stack.run_stack(batch, feats, params, dev, use_type_layer=True, path=path)
            _check_stack(got, want64, cfg.T * cfg.L, tol=TOL_INTERNAL, what="C1 np64 pad=%s path %d" % (pad, path))
            _check_stack(got, want, cfg.T * cfg.L, what="C1 torch pad=%s path %d" % (pad, path))


def test_module_path_graph_replay_is_bit_identical_to_eager(dev):
    """``ReasonGNNLayer._forward_stack`` with GNNRAG_GRAPH=1 (B * N < 4096: iteration 1 eager,
    iteration 2 captures the projection-free sequence, iterations 2..T replay it) against the eager module path on
    BASELINE config 1 - every layer's scores, distributions and node states bit for bit, over two consecutive forwards
    (a new batch binds a new stack: capture per forward, as main.py would run it)."""
    from gnnrag_amd import stack, synth
    cfg = synth.CONFIGS["C1"]
    assert cfg.B * cfg.N < 4096 and cfg.T >= 3
    batch, feats, params = synth.make_batch(cfg), synth.make_features(cfg), synth.make_layer_params(cfg)
    dvi = stack.DeviceInputs(batch, feats, dev)
    recs = {}
    for mode in ("eager", "graph"):
        layer = stack.build_layer(cfg, batch, params, dev)
        layer.graph_small = mode == "graph"
        for rep in range(2):
            stack.init_reason(layer, batch, dvi, dvi.h0)
            _, rec = stack.run_layers(layer, cfg, dvi, record=True)
            recs[(mode, rep)] = rec
        if mode == "graph":
            assert layer._stack._graph_rest is not None              # the replay really ran
    for rep in range(2):
        for key in ("score", "dist", "h"):
            for c in range(cfg.T * cfg.L):
                assert np.array_equal(recs[("eager", rep)][key][c], recs[("graph", rep)][key][c]), (key, c, rep)
