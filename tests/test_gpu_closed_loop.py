"""Closed loop on the GPU (judge's row n1): the reasoning part of ``ReaRev.forward`` and the ``Evaluator`` tail run
end to end on the HIP path - TypeLayer -> T x (L ReasonGNNLayer calls + QueryReform) -> candidate selection - with
the batch tuple, the state_dict and the encoder outputs as the ONLY inputs; nothing recorded is fed back between the
calls, so rounding differences propagate through every iteration exactly as in a real run.  Expected results come
from the LIVE reference's ``Evaluator.evaluate`` on CPU (tests/golden/make_golden_e2e.py ->
rearev_closed_loop.npz: 10 questions, 3 batches, entity_dim 50, num_iter 3, num_gnn 3, num_ins 2).

Asserted: ``pred`` (the Hits@1 decision) identical; ``pred_dist`` within the stated 1e-4; per question the SAME
retrieved candidates in the SAME order, identical precision / recall / F1 / Hits / EM, and an ``.info`` line equal to
the reference's in every field except the candidates' probabilities, which agree to 1e-4 (they are fp32 values
printed with 17 digits: byte equality would need bit-identical floats across CPU and GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL_STATED = 1e-4


def _modules(z, dev):
    from gnnrag_amd.modules.kg_reasoning.reasongnn import ReasonGNNLayer
    from gnnrag_amd.modules.layer_init import TypeLayer
    from gnnrag_amd.modules.query_update import QueryReform
    D, I, L = int(z["D"]), int(z["I"]), int(z["L"])
    args = dict(use_cuda=True, normalized_gnn=bool(int(z["normalized_gnn"])), num_ins=I, num_gnn=L,
                pos_emb=bool(int(z["pos_emb"])), linear_dropout=0.0)
    sd = lambda prefix: {k[len("param." + prefix):]: torch.from_numpy(z[k]) for k in z.files
                         if k.startswith("param." + prefix)}
    reasoning = ReasonGNNLayer(args, int(z["num_entity"]), int(z["num_kb_relation"]), D, "bfs")
    reasoning.load_state_dict(sd("reasoning."), strict=True)
    tl = TypeLayer(D, D, torch.nn.Dropout(0.0), dev, bool(int(z["norm_rel"])))
    tl.load_state_dict(sd("type_layer."), strict=True)
    reforms = []
    for j in range(I):
        r = QueryReform(D)
        r.load_state_dict(sd("reform%d." % j), strict=True)
        reforms.append(r.to(dev).eval())
    return tl.to(dev).eval(), reasoning.to(dev).eval(), reforms


@pytest.mark.parametrize("use_stack", [True, False], ids=["whole-iteration calls", "per-layer calls"])
def test_closed_loop_evaluation_matches_live_reference(use_stack):
    import gnnrag_amd  # noqa: F401
    import oracle.eval_tail as oe
    from gnnrag_amd import _lib, eval_tail, stack
    _lib.load()
    dev = torch.device("cuda", 0)
    z = np.load(os.path.join(GOLDEN, "rearev_closed_loop.npz"))
    tl, reasoning, reforms = _modules(z, dev)
    reasoning.use_stack = use_stack
    T, N = int(z["T"]), int(z["max_local_entity"])
    eps = float(z["eps"])
    ignore_prob = (1 - eps) / N                                                  # evaluate.py:156
    id2entity = {i: str(s) for i, s in enumerate(z["id2entity"])}
    pad = len(id2entity)
    info = [json.loads(str(l)) for l in z["info"]]
    qi = 0
    for k in range(int(z["n_batches"])):
        g = lambda name: z["b%d.%s" % (k, name)]
        F = len(g("heads"))
        et = (g("heads"), g("rels"), g("tails"), g("batch_ids"), np.arange(F), g("weight_list").tolist(),
              g("weight_rel_list").tolist())
        t = lambda a, dt=torch.float32: torch.from_numpy(np.asarray(a)).to(dev, dt)
        local_entity = torch.from_numpy(g("local_entity")).to(dev)
        pred, pred_dist = stack.run_rearev_loop(
            tl, reasoning, reforms, local_entity=local_entity, query_entities=t(g("query_entities")), edge_tuple=et,
            seed_dist=t(g("seed_dist")), rel_features=t(g("rel_features")), rel_features_inv=t(g("rel_features_inv")),
            instructions=t(g("ins0")), num_iter=T)
        assert len(reasoning.possible_cand) == int(g("calls"))                   # T x L layer calls, like the reference
        assert np.array_equal(pred.cpu().numpy(), g("pred"))                     # Hits@1 decisions
        err = np.abs(pred_dist.cpu().numpy() - g("pred_dist")).max()
        assert err <= TOL_STATED, (k, err)
        # Evaluator tail on the device, metrics by the restated f1_and_hits (pinned on CPU against these very lines)
        picked = eval_tail.retrieved_candidates(pred_dist, g("local_entity"), g("query_entities"), pad, ignore_prob, eps)
        answers = json.loads(str(g("answers")))
        for b, (cand2prob, _) in enumerate(picked):
            want = info[qi]
            got = oe.info_record(want["question"], T, answers[b], cand2prob, id2entity, None, eps)
            got = json.loads(json.dumps(got))                                    # what the writer would have written
            assert set(got) == set(want)
            for key in want:
                if key != "cand":
                    assert got[key] == want[key], (qi, key, got[key], want[key])
            assert [c for c, _ in got["cand"]] == [c for c, _ in want["cand"]], qi      # same entities, same order
            dp = max((abs(a[1] - b_[1]) for a, b_ in zip(got["cand"], want["cand"])), default=0.0)
            assert dp <= TOL_STATED, (qi, dp)
            qi += 1
    assert qi == len(info)
