"""NSM layer drop-in: the float64 restatement against outputs and gradients of
the live reference's NSMLayer (CPU), and the HIP-backed module against the same fixture (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

# tag -> (reason_kb, normalized_gnn, direction): NSMLayer walks head -> tail, NSMLayer_back tail -> head
CASES = {"plain": (False, False, 0), "kb_norm": (True, True, 0), "back_plain": (False, False, 1),
         "back_kb_norm": (True, True, 1)}


def _load():
    z = np.load(os.path.join(GOLDEN, "nsm_layer.npz"))
    F = len(z["heads"])
    et = (z["heads"], z["rels"], z["tails"], z["heads"] // int(z["N"]), np.arange(F, dtype=np.int64),
          z["weight_list"].tolist(), [1.0] * F)
    params = {k[6:]: z[k] for k in z.files if k.startswith("param.")}
    return z, et, params


@pytest.mark.parametrize("tag", list(CASES))
def test_nsm_oracle_matches_reference(tag):
    import oracle.nsm_layer as on
    z, et, params = _load()
    reason_kb, normalized, direction = CASES[tag]
    relfeat = z["rel_features_inv"] if direction else z["rel_features"]
    got = on.run(et, int(z["B"]), int(z["N"]), z["local_entity"], int(z["num_entity"]), z["h0"], relfeat,
                 z["ins"], z["seed_dist"], params, reason_kb=reason_kb, normalized_gnn=normalized, direction=direction,
                 Gd=z["Gd"], Gh=z["Gh"])
    for c in range(int(z["L"])):
        assert np.abs(got["dist"][c] - z[tag + ".ref.dist"][c]).max() <= 2e-6
        assert np.abs(got["h"][c] - z[tag + ".ref.h"][c]).max() <= 2e-5
    for k in [f[len(tag) + 6:] for f in z.files if f.startswith(tag + ".grad.")]:
        want = z["%s.grad.%s" % (tag, k)]
        np.testing.assert_allclose(got["grad"][k], want, rtol=0, atol=3e-4 * max(np.abs(want).max(), 1e-3), err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("grad", [False, True], ids=["inference", "autograd"])
@pytest.mark.parametrize("tag", list(CASES))
def test_nsm_module_matches_reference_fixture(tag, grad):
    import gnnrag_amd  # noqa: F401
    from gnnrag_amd.modules.kg_reasoning.nsm_gnn import NSMLayer, NSMLayer_back
    dev = torch.device("cuda", 0)
    z, et, params = _load()
    reason_kb, normalized, direction = CASES[tag]
    B, N, D, L = int(z["B"]), int(z["N"]), int(z["D"]), int(z["L"])
    args = dict(use_cuda=True, normalized_gnn=normalized, num_step=L, reason_kb=reason_kb, linear_dropout=0.0)
    layer = (NSMLayer_back if direction else NSMLayer)(args, int(z["num_entity"]), int(z["R1"]) - 1, D)
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    layer = layer.to(dev)
    layer.train(grad)
    X = {k: torch.tensor(z[k], device=dev, requires_grad=grad) for k in ("h0", "ins")}
    X["rel_features"] = torch.tensor(z["rel_features_inv"] if direction else z["rel_features"], device=dev,
                                     requires_grad=grad)
    with torch.set_grad_enabled(grad):
        layer.init_reason(local_entity=torch.from_numpy(z["local_entity"]).to(dev), kb_adj_mat=et,
                          local_entity_emb=X["h0"], rel_features=X["rel_features"])
        if direction:
            layer.rel_features_inv = X["rel_features"]          # as the reference's caller has to (nsm_gnn.py:122)
        dist = torch.from_numpy(z["seed_dist"]).float().to(dev)
        loss = 0.0
        for j in range(L):
            score, dist = layer(dist, X["ins"][j], step=j, return_score=True)
            assert np.abs(dist.detach().cpu().numpy() - z[tag + ".ref.dist"][j]).max() <= 1e-4
            assert np.abs(layer.local_entity_emb.detach().cpu().numpy() - z[tag + ".ref.h"][j]).max() <= 1e-4
            assert (dist.argmax(1).cpu().numpy() == z[tag + ".ref.dist"][j].argmax(1)).all()
            np.testing.assert_array_equal(layer.possible_cand[j].cpu().numpy(), z[tag + ".possible_cand"][j])
            if grad:
                loss = loss + (dist * torch.from_numpy(z["Gd"][j]).to(dev)).sum()
        if grad:
            loss = loss + (layer.local_entity_emb * torch.from_numpy(z["Gh"]).to(dev)).sum()
            loss.backward()
            got = {k: v.grad for k, v in X.items()}
            got.update({k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
            for k in [f[len(tag) + 6:] for f in z.files if f.startswith(tag + ".grad.")]:
                want = z["%s.grad.%s" % (tag, k)]
                np.testing.assert_allclose(got[k].cpu().numpy(), want, rtol=0,
                                           atol=3e-4 * max(np.abs(want).max(), 1e-3), err_msg=k)
