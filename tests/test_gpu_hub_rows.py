"""Hub rows of the gather walk (tables larger than LDS; BASELINE config 5): the rows are stored in relation order and
walked either as a dense hub-by-relation product (k_hub_weights / k_hub_dense / k_hub_finish, the default) or in
256-fact chunks with one table row per run of equal relations (k_heavy_partial, GNNRAG_HUB_DENSE=0).  Both against a
float64 sum over the caller's fact tuple, on a graph built to hit the corner cases: a run of one relation over a dozen
chunks, hubs of exactly 256 k facts, a hub of 257 facts with 257 relations, a hub in the forward direction, a question
without hubs.  Reference semantics: fact2tail . (fact_val * fact_prior) of reasongnn.py:80-116 in the factored
(per-question relation table) form."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(seed=3):
    rng = np.random.default_rng(seed)
    B, N, R = 3, 3000, 2200
    H, Rl, T = [], [], []

    def add(q, h, r, t):
        H.append(np.asarray(h) + q * N); Rl.append(np.asarray(r)); T.append(np.asarray(t) + q * N)

    for q in range(B):                                     # background: every relation in use, light rows
        add(q, rng.integers(20, N, 9000), rng.integers(0, R, 9000), rng.integers(20, N, 9000))
        add(q, np.arange(20, 20 + R), np.arange(R), rng.integers(20, N, R))
    # question 0, node 5 as head (direction 1 hub): 3000 facts of relation 7 (12 chunks of one run), 2500 spread
    add(0, np.full(3000, 5), np.full(3000, 7), rng.integers(20, N, 3000))
    add(0, np.full(2500, 5), rng.integers(0, R, 2500), rng.integers(20, N, 2500))
    # node 6: exactly 512 facts over 3 relations; node 7: 257 facts, 257 relations; node 8: exactly 256 (not a hub)
    add(0, np.full(512, 6), rng.integers(100, 103, 512), rng.integers(20, N, 512))
    add(0, np.full(257, 7), np.arange(300, 557), rng.integers(20, N, 257))
    add(0, np.full(256, 8), rng.integers(0, R, 256), rng.integers(20, N, 256))
    # question 1: a hub as TAIL (direction 0) whose relation run ends exactly at a 64-fact batch boundary
    add(1, rng.integers(20, N, 64), np.full(64, 11), np.full(64, 9))
    add(1, rng.integers(20, N, 64 * 5), np.repeat(np.arange(12, 17), 64), np.full(64 * 5, 9))
    add(1, rng.integers(20, N, 700), rng.integers(0, R, 700), np.full(700, 9))
    # question 2: no hubs
    h, r, t = (np.concatenate(x).astype(np.int64) for x in (H, Rl, T))
    p = rng.permutation(len(h))
    return B, N, R, h[p], r[p], t[p]


def _graph_blocks_do_not_fit(seed=4):
    """Many hubs of few facts over a large relation vocabulary: the hub-by-relation weight blocks (hubs x relations x
    4 B) outgrow the workspace region they live in (16 B per fact), so the device-side decision must be the chunked
    fallback."""
    rng = np.random.default_rng(seed)
    B, N, R = 2, 3000, 2200
    H, Rl, T = [], [], []
    for q in range(B):
        H.append(np.arange(100, 100 + R) + q * N); Rl.append(np.arange(R)); T.append(rng.integers(100, N, R) + q * N)
        for hub in range(90):                                 # 90 hubs x 260 facts each, as heads (direction 1)
            H.append(np.full(260, hub) + q * N); Rl.append(rng.integers(0, R, 260)); T.append(rng.integers(100, N, 260) + q * N)
    h, r, t = (np.concatenate(x).astype(np.int64) for x in (H, Rl, T))
    p = rng.permutation(len(h))
    return B, N, R, h[p], r[p], t[p]


def _run(dev, graph=None, want_form=None, D=200, weights=False):
    from gnnrag_amd import ops
    B, N, R, h, r, t = (graph or _graph)()
    plan = ops.CsrPlan(h, r, t, B, N, R, dev)
    assert ops.aggregate_fused_variant(plan, D) == ops.WALK_L2_GATHER
    got_plan = plan.to_host()
    if graph is None:
        assert got_plan["n_heavy"][0] >= 1 and got_plan["n_heavy"][1] >= 3
    # which hub form THIS call takes: the kernels' own predicate on the call's arguments, read back from the device
    form = ops.aggregate_fused_hub_form(plan, D, 1)
    if want_form is None:
        want_form = ops.HUB_FORM_NONE if os.environ.get("GNNRAG_HUB_DENSE") == "0" else ops.HUB_FORM_DENSE
    assert form["form"] == want_form and form["hubs"] == (got_plan["n_heavy"][0], got_plan["n_heavy"][1]), form
    rng = np.random.default_rng(9)
    wfact = None
    if weights:                                    # normalized_gnn: per-fact weights, used squared (base_gnn.py:38-47)
        wfact = (0.5 + rng.random(len(h))).astype(np.float32)
        plan.attach_w_gnn(wfact)
    dist = rng.random((B, N)).astype(np.float32)
    dist[:, ::3] = 0.0
    P = (rng.standard_normal((2, plan.rel_total, D)) * 0.3).astype(np.float32)
    out = ops.aggregate_fused(plan, torch.from_numpy(dist).to(dev), torch.from_numpy(P).to(dev)).cpu().numpy().reshape(B * N, D)
    # float64 reference from the caller's tuple
    rel_off = got_plan["rel_off"].astype(np.int64)
    rows = got_plan["rel_rows"]
    q = h // N
    # compact row of (question, relation): position in the sorted distinct pair list
    key = rows[:, 0].astype(np.int64) * (R + 1) + rows[:, 1]
    row_of = np.searchsorted(key, q * (R + 1) + r)
    want = np.zeros((B * N, D))
    d64 = dist.reshape(-1).astype(np.float64)
    w2 = np.ones(len(h)) if wfact is None else wfact.astype(np.float64) ** 2
    np.add.at(want, t, (w2 * d64[h])[:, None] * P[0][row_of].astype(np.float64))
    np.add.at(want, h, (w2 * d64[t])[:, None] * P[1][row_of].astype(np.float64))
    scale = np.abs(want).max()
    return out, want, scale, rel_off


def test_hub_rows_dense_form():
    import gnnrag_amd  # noqa: F401
    dev = torch.device("cuda", 0)
    out, want, scale, _ = _run(dev)         # asserts HUB_FORM_DENSE as read back from the device
    assert np.abs(out - want).max() <= 2e-5 * scale, np.abs(out - want).max() / scale


@pytest.mark.parametrize("D", [132, 160, 192, 196, 256])
def test_light_rows_four_facts_per_step_at_other_widths(D):
    """k_walk_light_q (round 4: four facts per step, a lane covers the columns 64 c + 4 l) with three column pieces
    (128 < D <= 192) and four (D <= 256), a last piece of a single float4 (132, 196), full pieces (192, 256); the same
    graph: light rows of 1 .. 256 facts (four 64-fact batches), hub rows beside them."""
    import gnnrag_amd  # noqa: F401
    out, want, scale, _ = _run(torch.device("cuda", 0), D=D)
    assert np.abs(out - want).max() <= 2e-5 * scale, np.abs(out - want).max() / scale


@pytest.mark.parametrize("D", [200, 160])
def test_gather_walk_with_per_fact_weights(D):
    """normalized_gnn weights on the gather walk: the light rows then take the run-per-direction form of k_walk_light_q
    (the merged record stream carries no weights), the hub rows their weighted sums."""
    import gnnrag_amd  # noqa: F401
    out, want, scale, _ = _run(torch.device("cuda", 0), D=D, weights=True)
    assert np.abs(out - want).max() <= 2e-5 * scale, np.abs(out - want).max() / scale


def test_hub_rows_device_side_fallback_when_the_weight_blocks_do_not_fit():
    """The silent device-side fallback, made visible: same entry point, dense form compiled in and enabled, but the
    weight blocks exceed the workspace - the read-back says CHUNKED and the result still matches float64."""
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import ops
    if os.environ.get("GNNRAG_HUB_DENSE") == "0":
        pytest.skip("dense hub form switched off in this process")
    out, want, scale, _ = _run(torch.device("cuda", 0), _graph_blocks_do_not_fit, ops.HUB_FORM_CHUNKED)
    assert np.abs(out - want).max() <= 2e-5 * scale, np.abs(out - want).max() / scale


def test_hub_rows_chunk_form_in_a_fresh_process():
    """The chunked form is chosen by the environment when the library is first used: a process of its own."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, torch, gnnrag_amd\n"
            "from test_gpu_hub_rows import _run\n"
            "out, want, scale, _ = _run(torch.device('cuda', 0))\n"
            "err = np.abs(out - want).max() / scale\n"
            "assert err <= 2e-5, err\n"
            "print('CHUNK_OK', err)\n") % (REPO, os.path.join(REPO, "tests"))
    env = dict(os.environ, GNNRAG_HUB_DENSE="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHUNK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_hub_rows_are_stored_in_relation_order():
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import ops
    dev = torch.device("cuda", 0)
    B, N, R, h, r, t = _graph()
    plan = ops.CsrPlan(h, r, t, B, N, R, dev)
    got = plan.to_host()
    for d, dst in ((0, t), (1, h)):
        rp = got["row_ptr%d" % d]
        e = got["edge%d" % d]
        perm = got["perm%d" % d]
        for n in got["heavy%d" % d]:
            rel = e[rp[n]:rp[n + 1], 1]
            assert (np.diff(rel) >= 0).all()                                  # relation order
            same = np.diff(rel) == 0
            assert (np.diff(perm[rp[n]:rp[n + 1]])[same] > 0).all()           # fact order inside a relation
            assert (dst[perm[rp[n]:rp[n + 1]]] == n).all()
        light = np.setdiff1d(np.flatnonzero(np.diff(rp) > 1), got["heavy%d" % d])[:200]
        for n in light:
            assert (np.diff(perm[rp[n]:rp[n + 1]]) > 0).all()                 # light rows keep the fact order
