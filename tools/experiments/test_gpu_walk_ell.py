"""The lane-per-node form of the fused LDS walk (round 5; csrc/csr_plan.hip k_ell_plan, csrc/aggregate.hip k_fact_prior_ell /
k_walk_ell): the structure carries a question's light nodes sorted by the length of their merged run in sets of 64, a wave
walks a set with one lane per node.  Checked here: the sets against a numpy restatement (every non-big node exactly once,
longest runs first, node id ascending among equals; every slot holds the lane's step-th merged record or padding; the
padding bound that sizes the structure), the walk against the float64 sum over the caller's fact tuple on graphs with
empty nodes, more big nodes than the kernel's LDS list holds, a hub, ragged N, several questions, per-fact weights - and
BIT-IDENTITY with the 4-lane-group form (GNNRAG_WALK_ELL=0 in a child process): a lane adds its node's facts in merged order,
the same sum as before.  Reference semantics: fact2tail . (fact_val * fact_prior) of reasongnn.py:80-116 in the factored form."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIG_DEG = 32


def _graph(seed, B, N, R, facts, n_medium=0, n_many_big=0, hub=0, empty_from=None):
    rng = np.random.default_rng(seed)
    H, Rl, T = [], [], []
    top = empty_from or N                                     # nodes >= empty_from have no facts at all
    for q in range(B):
        h = rng.integers(0, top, facts)
        t = (rng.zipf(1.7, facts) - 1) % top if q % 2 == 0 else rng.integers(0, top, facts)
        H.append(h + q * N); Rl.append(rng.integers(0, R, facts)); T.append(t + q * N)
        for k in range(n_medium):                             # medium rows (a whole wave per node): 100 .. 900 facts
            m = int(rng.integers(100, 900))
            H.append(rng.integers(0, top, m) + q * N); Rl.append(rng.integers(0, R, m)); T.append(np.full(m, 3 + k) + q * N)
        for k in range(n_many_big):                           # more big nodes than the LDS list (72): 40 facts each
            H.append(np.full(40, 200 + k) + q * N); Rl.append(rng.integers(0, R, 40)); T.append(rng.integers(0, top, 40) + q * N)
        if hub and q == 0:                                    # one row of > 4096 facts (whole workgroup)
            H.append(rng.integers(0, top, hub)); Rl.append(rng.integers(0, R, hub)); T.append(np.full(hub, 1))
    h, r, t = (np.concatenate(x).astype(np.int64) for x in (H, Rl, T))
    p = rng.permutation(len(h))
    return h[p], r[p], t[p]


# (more than ~310 relations in use per question: the tables then take the 16-column slices the lane-per-node kernel serves)
CASES = {
    "plain": dict(B=3, N=700, R=500, facts=4000),
    "ragged_empty": dict(B=2, N=1001, R=450, facts=3000, empty_from=600),
    "medium_and_hub": dict(B=2, N=900, R=600, facts=5000, n_medium=6, hub=5000),
    "many_big": dict(B=2, N=1200, R=500, facts=3000, n_many_big=90),
    "one_question": dict(B=1, N=2000, R=600, facts=10000, n_medium=3),
    "nine_questions": dict(B=9, N=320, R=500, facts=2500),
}


def _ell_reference(rp0, rp1, edge_m, m_from, B, N):
    """numpy restatement of k_ell_plan: (nsets [B], sets [*, 2], nodes [*, 64], records [*, 3])."""
    nsets, sets, nodes, recs = [], [], [], []
    for g in range(B):
        n = np.arange(g * N, (g + 1) * N)
        l0, l1 = rp0[n + 1] - rp0[n], rp1[n + 1] - rp1[n]
        light = np.maximum(l0, l1) <= BIG_DEG
        ln = (l0 + l1)[light]
        ids = n[light]
        order = np.lexsort((ids, -ln))                        # longest first, node id ascending among equals
        ids, ln = ids[order], ln[order]
        slot = int(rp0[g * N] + rp1[g * N]) + 4096 * g
        ns = (len(ids) + 63) // 64
        nsets.append(ns)
        for s in range(ns):
            sid = ids[64 * s: 64 * s + 64]
            sl = ln[64 * s: 64 * s + 64]
            steps = int(sl[0])
            sets.append((slot, steps))
            nodes.append(np.concatenate([sid, np.full(64 - len(sid), -1)]))
            for j in range(steps):
                for lane in range(64):
                    if lane < len(sid) and j < sl[lane]:
                        m = int(rp0[sid[lane]] + rp1[sid[lane]]) + j
                        recs.append((edge_m[m, 0], edge_m[m, 1], m_from[m]))
                    else:
                        recs.append((-1, 0, 0))
            slot += 64 * steps
    return (np.array(nsets), np.array(sets).reshape(-1, 2), np.array(nodes).reshape(-1, 64),
            np.array(recs, dtype=np.int64).reshape(-1, 3))


@pytest.fixture(scope="module")
def dev():
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)


@pytest.mark.parametrize("case", ["plain", "ragged_empty", "medium_and_hub", "many_big", "nine_questions"])
def test_sets_of_the_structure_against_numpy(dev, case):
    from gnnrag_amd import ops
    c = CASES[case]
    h, r, t = _graph(11, **c)
    plan = ops.CsrPlan(h, r, t, c["B"], c["N"], c["R"] + 2, dev)
    got = plan.to_host()
    ns, sets, nodes, recs = _ell_reference(got["row_ptr0"].astype(np.int64), got["row_ptr1"].astype(np.int64),
                                           got["edge_m"], got["m_from"], c["B"], c["N"])
    np.testing.assert_array_equal(got["ell_nsets"], ns)
    np.testing.assert_array_equal(got["ell_sets"], sets)
    np.testing.assert_array_equal(got["ell_node"], nodes)
    np.testing.assert_array_equal(got["ell_records"], recs)
    # every non-big node exactly once; the padding bound behind the structure's size (<= 64 x 64 slots per question)
    big = set(int(x) for b in got["big"] for x in b)
    listed = nodes[nodes >= 0]
    assert len(np.unique(listed)) == len(listed) == c["B"] * c["N"] - len(big) and not (set(listed.tolist()) & big)
    first = 0
    for g in range(c["B"]):
        s = sets[first: first + ns[g]]
        first += ns[g]
        used = int((64 * s[:, 1]).sum())
        assert used <= 2 * len(h) + 4096 and (ns[g] == 0 or (np.diff(s[:, 1]) <= 0).all())


def _fused_inputs(c, seed, dev, weights=False):
    from gnnrag_amd import ops
    h, r, t = _graph(seed, **c)
    B, N, R1, D = c["B"], c["N"], c["R"] + 2, 200
    plan = ops.CsrPlan(h, r, t, B, N, R1, dev)
    rng = np.random.default_rng(seed + 1)
    P = (0.3 * rng.standard_normal((2, plan.rel_total, D))).astype(np.float32)
    dist = rng.random((B, N)).astype(np.float32)
    dist[:, ::7] = 0.0                                        # exact zeros among the priors
    w = None
    if weights:
        w = (0.5 + rng.random(len(h))).astype(np.float32)
        plan.attach_w_gnn(w)
    return plan, h, r, t, P, dist, w


def _float64_sum(plan, h, r, t, P, dist, w, B, N):
    rows = plan.rel_rows()
    key = {(int(b), int(rr)): i for i, (b, rr) in enumerate(rows)}
    want = np.zeros((B * N, P.shape[2]))
    d64, P64 = dist.reshape(-1).astype(np.float64), P.astype(np.float64)
    for f in range(len(h)):
        row = key[(int(h[f]) // N, int(r[f]))]
        wf = float(w[f]) ** 2 if w is not None else 1.0       # normalized_gnn: the weight enters twice (reasongnn.py:80,84)
        want[t[f]] += wf * d64[h[f]] * P64[0, row]
        want[h[f]] += wf * d64[t[f]] * P64[1, row]
    return want


@pytest.mark.parametrize("case", list(CASES))
def test_lane_per_node_walk_against_float64(dev, case):
    from gnnrag_amd import ops
    c = CASES[case]
    plan, h, r, t, P, dist, w = _fused_inputs(c, 21, dev, weights=(case == "plain"))
    assert ops.aggregate_fused_variant(plan, 200) == ops.WALK_LDS_16
    got = ops.aggregate_fused(plan, torch.from_numpy(dist).to(dev), torch.from_numpy(P).to(dev)).cpu().numpy()
    want = _float64_sum(plan, h, r, t, P, dist, w, c["B"], c["N"])
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    again = ops.aggregate_fused(plan, torch.from_numpy(dist).to(dev), torch.from_numpy(P).to(dev)).cpu().numpy()
    assert np.array_equal(got, again)


CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import gnnrag_amd
from gnnrag_amd import ops
import test_gpu_walk_ell as T
dev = torch.device("cuda", 0)
out = {}
for case in T.CASES:
    plan, h, r, t, P, dist, w = T._fused_inputs(T.CASES[case], 21, dev, weights=(case == "plain"))
    out[case] = ops.aggregate_fused(plan, torch.from_numpy(dist).to(dev), torch.from_numpy(P).to(dev)).cpu().numpy()
    seed = np.zeros_like(dist); seed[:, 5] = 1.0
    out[case + "_seed"] = ops.aggregate_fused(plan, torch.from_numpy(seed).to(dev), torch.from_numpy(P).to(dev)).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_bit_identical_to_the_lane_group_form(dev, tmp_path):
    outs = {}
    for flag in ("1", "0"):
        path = str(tmp_path / ("ell%s.npz" % flag))
        env = dict(os.environ, GNNRAG_WALK_ELL=flag)
        r = subprocess.run([sys.executable, "-c", CHILD % (REPO, REPO), path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[flag] = np.load(path)
    for k in outs["1"].files:
        a, b = outs["1"][k], outs["0"][k]
        if k.startswith("many_big"):
            # more big nodes than the kernels' LDS list holds: k_walk_slice then walks EVERY node of the question by its
            # lane group (one chain per row), k_walk_ell still gives a big row to a whole wave (lane partial sums + a fixed
            # tree) - two fixed orders of the same sum
            assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max())
        else:
            assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))
