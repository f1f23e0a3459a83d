"""Oracle parity of the HIP path AT THE SHAPES OF THE BASELINE CONFIGS (BASELINE.json configs C1, C2, C4, C5),
each test naming the aggregation kernel variant it proves (``ops.aggregate_fused_variant`` = the host-side
dispatch decision of ``gnnrag_aggregate_fused``).  Per-question shapes are the configs' own; the batch is cut to
what the CPU restatement (``oracle/rearev_torch_cpu.py``, pinned to the live reference by tests/golden) finishes
in seconds - questions are independent subgraphs, so a question's result does not depend on the batch around it
(checked bit for bit at full C2 size in test_gpu_parity.py::test_full_size_properties_c2).

Tolerance: north_star's 1e-4 fp32 on node embeddings / distributions, argmax (= Hits@1 decision) identical.
Reference path: reasongnn.py:61-174, base_gnn.py:19-51."""
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
        # node embeddings: fp32 rounding scales with their magnitude (a TypeLayer start sums hundreds of facts at the
        # hubs); distributions are <= 1
        eh = np.abs(got["h"][c] - want["h"][c]).max() / max(1.0, np.abs(want["h"][c]).max())
        ed = np.abs(got["dist"][c] - want["dist"][c]).max()
        assert eh <= tol and ed <= tol, (what, c, eh, ed)
        assert (got["dist"][c].argmax(1) == want["dist"][c].argmax(1)).all(), (what, c)


def _slice_questions(batch, feats, lo, hi):
    """Questions [lo, hi) of a batch as a self-contained batch (what a rank would own)."""
    from gnnrag_amd import shard, synth
    cfg = batch.cfg
    et = shard.shard_edge_tuple(batch.edge_tuple, cfg.N, lo, hi)
    sub = synth.Batch(cfg=synth.GraphConfig(**{**cfg.__dict__, "B": hi - lo}), local_entity=batch.local_entity[lo:hi],
                      query_entities=batch.query_entities[lo:hi], seed_dist=batch.seed_dist[lo:hi], edge_tuple=et,
                      num_entity=batch.num_entity, n_real=batch.n_real[lo:hi])
    sfe = dict(feats)
    sfe["h0"] = feats["h0"][lo:hi]
    sfe["ins"] = feats["ins"][:, lo:hi]
    return sub, sfe


def test_c5_shape_tables_larger_than_lds(dev):
    """C5 questions (20 000 nodes, 200 000 typed edges, all 6 000 relation types in use): the per-question
    relation tables (2 x 6001 rows) exceed a CU's LDS, so the fused aggregation takes the table-row gather
    kernel with float4 lanes and the XCD-aware block mapping - the kernel a full C5 batch runs (same rel_max,
    same D => same dispatch)."""
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import ops, stack, synth
    c5 = synth.CONFIGS["C5"]
    cfg = synth.GraphConfig(**{**c5.__dict__, "name": "C5x2", "B": 2, "seed": 505})
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    plan = _plan_of(batch, dev)
    assert plan.rel_max > 1240                                        # more rows than fit 160 KB as 16-column slices
    full = _lib_csr_like(plan, B=c5.B)
    assert ops.aggregate_fused_variant(plan, cfg.D) == full == ops.WALK_L2_GATHER
    assert cfg.N % 4 == 0 and cfg.D % 4 == 0                          # float4 lanes, bpg (XCD-aware) mapping active
    want = otorch.run_stack(batch, feats, params)
    for path in (0, 2, 1):                                            # auto (= fused at this shape), fused, unfused
        got = stack.run_stack(batch, feats, params, dev, path=path)
        _check_stack(got, want, cfg.T * cfg.L, what="C5 path %d" % path)


def _lib_csr_like(plan, B):
    """Dispatch decision for a batch of B questions with the same per-question shape (rel_max) as `plan`."""
    import ctypes
    from gnnrag_amd import _lib
    c = _lib.CsrStruct()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(plan.c), ctypes.sizeof(c))
    c.B = B
    c.rel_total = plan.rel_max * B
    return _lib.load().gnnrag_aggregate_fused_variant(ctypes.byref(c), 200)


def test_c4_shape_three_instructions_four_layers(dev):
    """C4 questions (5 000 nodes, 30 000 typed edges, num_ins 3, 4 layers): the 16-column LDS walk with its
    big-node list in use (tens of hub nodes per question, below the list's capacity), the three-instruction
    unfused walk and the K = 7D update GEMM."""
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import ops, stack, synth
    c4 = synth.CONFIGS["C4"]
    cfg = synth.GraphConfig(**{**c4.__dict__, "name": "C4x2", "B": 2, "seed": 404})
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    plan = _plan_of(batch, dev)
    assert ops.aggregate_fused_variant(plan, cfg.D) == ops.WALK_LDS_16
    nbig = [len(b) for b in plan.to_host()["big"]]
    assert all(0 < n <= 72 for n in nbig), nbig                       # listed hubs: wave / workgroup classes run
    want = otorch.run_stack(batch, feats, params)
    for path in (2, 1):
        got = stack.run_stack(batch, feats, params, dev, path=path)
        _check_stack(got, want, cfg.T * cfg.L, what="C4 path %d" % path)


def test_more_big_nodes_than_the_list_holds(dev):
    """Dense questions in which most nodes have more than 32 facts: the LDS walk's per-question hub list
    overflows (> 72 entries) and every row is walked by its 4-lane group instead."""
    import oracle.rearev_np64 as onp
    from gnnrag_amd import ops, stack, synth
    cfg = synth.GraphConfig(name="dense", B=3, N=500, E=24000, R=50, D=200, I=2, L=2, T=1, seed=77,
                            zipf_heads=False)
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    plan = _plan_of(batch, dev)
    assert ops.aggregate_fused_variant(plan, cfg.D) in (ops.WALK_LDS_16, ops.WALK_LDS_32)
    nbig = [len(b) for b in plan.to_host()["big"]]
    assert all(n > 72 for n in nbig), nbig
    want = onp.run_stack(batch, feats, params)
    got = stack.run_stack(batch, feats, params, dev, path=2)
    _check_stack(got, want, cfg.T * cfg.L, tol=TOL_INTERNAL, what="dense")


def test_c1_shape_single_question(dev):
    """C1: one WebQSP-shaped question (B = 1, N = 2000 padded, released-checkpoint dims D = 50, 2 instructions,
    3 layers, 3 iterations) incl. TypeLayer.  D = 50 is not a multiple of 4: the drop-in module zero-pads it to 56
    (reasongnn.py drop-in, `_inference_params`), so what runs is the LDS walk at width 56 (k_walk_slice) and the
    float4 GEMM variants - asserted below on the PADDED width.  (The unpadded scalar path, GNNRAG_PAD_DIM=0, is
    covered by tests/test_gpu_round3_shapes.py::test_c1_kernel_variants_padded_and_unpadded.)"""
    import oracle.rearev_np64 as onp
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import ops, stack, synth
    cfg = synth.CONFIGS["C1"]
    for seed in (cfg.seed, 12, 13):
        batch = synth.make_batch(cfg, seed=seed)
        feats = synth.make_features(cfg, seed=seed)
        params = synth.make_layer_params(cfg)
        plan = _plan_of(batch, dev)
        assert cfg.D % 4 == 2 and ops.aggregate_fused_variant(plan, 56) in (ops.WALK_LDS_16, ops.WALK_LDS_32)
        want64 = onp.run_stack(batch, feats, params, use_type_layer=True)
        want = otorch.run_stack(batch, feats, params, use_type_layer=True)
        for path in (0, 1, 2):
            got = stack.run_stack(batch, feats, params, dev, use_type_layer=True, path=path)
            # TypeLayer sums hundreds of facts at the hubs (|h0| up to ~1e2): fp32 rounding scales with the magnitude
            assert np.abs(got["h0"] - want64["h0"]).max() <= TOL_INTERNAL * max(1.0, np.abs(want64["h0"]).max())
            _check_stack(got, want64, cfg.T * cfg.L, tol=TOL_INTERNAL, what="C1 np64 path %d" % path)
            _check_stack(got, want, cfg.T * cfg.L, what="C1 torch path %d" % path)


@pytest.fixture(params=["mixed", "bf16x3", "fp32"])
def math_mode(request):
    """C2 at full size is where the W-resident kernels run (k_gemm_wres / k_update_b3 / k_tables_b3 need >= 8192 rows):
    each math mode of the binding is pinned in turn."""
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import ops
    old = ops.set_dense_math({"mixed": ops.MATH_MIXED, "bf16x3": ops.MATH_BF16X3, "fp32": ops.MATH_FP32}[request.param])
    yield request.param
    ops.set_dense_math(old)


@pytest.mark.parametrize("path", [2, 1], ids=["fused", "unfused"])
def test_c2_full_batch_against_oracle_slices(dev, path, math_mode):
    """C2 at FULL size (B = 64, the batch bench.py times) on the GPU; the oracle runs on two 2-question slices
    of the SAME batch (first and last questions, i.e. both ends of the XCD / work-item mapping)."""
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import ops, stack, synth
    cfg = synth.CONFIGS["C2"]
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    plan = _plan_of(batch, dev)
    assert ops.aggregate_fused_variant(plan, cfg.D) == ops.WALK_LDS_16
    got = stack.run_stack(batch, feats, params, dev, path=path)
    for lo, hi in ((0, 2), (cfg.B - 2, cfg.B)):
        sub, sfe = _slice_questions(batch, feats, lo, hi)
        want = otorch.run_stack(sub, sfe, params)
        part = {k: [x[lo:hi] for x in got[k]] for k in ("h", "dist")}
        _check_stack(part, want, cfg.T * cfg.L, what="C2 questions %d:%d path %d" % (lo, hi, path))


def _full_batch_vs_oracle_slices(cfg, dev, path, slices, what):
    """The FULL batch of a config on the GPU (only the listed question slices are copied back per layer call) against
    the torch-CPU oracle on those slices of the same batch.  Returns the layer's structure."""
    import oracle.rearev_torch_cpu as otorch
    from gnnrag_amd import stack, synth
    batch = synth.make_batch(cfg)
    feats = synth.make_features(cfg)
    params = synth.make_layer_params(cfg)
    dvi = stack.DeviceInputs(batch, feats, dev)
    layer = stack.build_layer(cfg, batch, params, dev, path)
    stack.init_reason(layer, batch, dvi, dvi.h0)
    got = {sl: {"h": [], "dist": []} for sl in slices}
    with torch.no_grad():
        for t in range(cfg.T):
            dist = dvi.seed_dist
            for j in range(cfg.L):
                _, dist = layer(dist, dvi.ins[t], step=j, return_score=True)
                for lo, hi in slices:
                    got[(lo, hi)]["h"].append(layer.local_entity_emb[lo:hi].cpu().numpy())
                    got[(lo, hi)]["dist"].append(dist[lo:hi].cpu().numpy())
    for lo, hi in slices:
        sub, sfe = _slice_questions(batch, feats, lo, hi)
        want = otorch.run_stack(sub, sfe, params)
        _check_stack(got[(lo, hi)], want, cfg.T * cfg.L, what="%s questions %d:%d path %d" % (what, lo, hi, path))
    return layer.plan


def test_c5_full_batch_dense_hub_form_against_oracle_slices(dev):
    """BASELINE config 5 at its FULL per-GPU batch (32 questions x 20 000 nodes, 7.04 M facts, every question uses all
    6001 relation rows): the gather walk with the XCD-aware mapping and - asserted by reading the kernels' own
    decision back from the device - the DENSE hub form (weight blocks sized on the device fit the workspace at B = 32,
    8 relation ranges per question: the path bench.py --workload C5 times).  Oracle: first and last question."""
    from gnnrag_amd import ops, synth
    cfg = synth.CONFIGS["C5"]
    assert cfg.B == 32
    plan = _full_batch_vs_oracle_slices(cfg, dev, 2, [(0, 1), (cfg.B - 1, cfg.B)], "C5 full")
    assert ops.aggregate_fused_variant(plan, cfg.D) == ops.WALK_L2_GATHER
    form = ops.aggregate_fused_hub_form(plan, cfg.D, cfg.I)
    assert form["form"] == ops.HUB_FORM_DENSE, form
    assert form["hubs"][1] >= cfg.B and form["relation_ranges"] == 8, form     # Zipf heads: hubs in the inverse direction


def test_c4_full_batch_against_oracle_slices(dev):
    """BASELINE config 4 at its FULL batch (32 questions x 5000 nodes, 30 000 typed edges, 3 instructions, 4 layers):
    the 16-column LDS walk (no dense hub kernels in that dispatch - asserted) and the three-instruction tables."""
    from gnnrag_amd import ops, synth
    cfg = synth.CONFIGS["C4"]
    assert cfg.B == 32
    for path in (2, 1):
        plan = _full_batch_vs_oracle_slices(cfg, dev, path, [(0, 2), (cfg.B - 2, cfg.B)], "C4 full")
    assert ops.aggregate_fused_variant(plan, cfg.D) == ops.WALK_LDS_16
    assert ops.aggregate_fused_hub_form(plan, cfg.D, cfg.I)["form"] == ops.HUB_FORM_NONE
