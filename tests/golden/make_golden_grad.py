#!/usr/bin/env python
"""Generates tests/golden/grad_*.npz from the LIVE reference (cmavro/GNN-RAG @ v2): gradients that
``torch.autograd`` derives through the reference's own ``ReasonGNNLayer`` / ``TypeLayer`` modules,
for the inputs already stored in ``layer_d200.npz`` / ``layer_d50.npz`` / ``typelayer.npz``.

Run in the build container only (needs /root/reference, CPU is enough):

    python tests/golden/make_golden_grad.py

Loss (fixed, stored cotangents):  sum_c <dist_c, Gd_c>  +  <h_last, Gh>   over the T*L layer calls c,
so every layer's distribution and the chained node states receive a gradient
(train_model.py:222-228 back-propagates a KL loss on the last distribution the same way).
Stored: ``grad.<parameter or input name>`` in float64->float32, ``cot.Gd`` / ``cot.Gh``.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/gnn"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REF)

from conftest import load_golden  # noqa: E402
from make_golden import layer_args  # noqa: E402


def layer_grads(name):
    from modules.kg_reasoning.reasongnn import ReasonGNNLayer
    cfg, batch, feats, params, ref = load_golden(name)
    rng = np.random.default_rng(7)
    ncall = cfg.T * cfg.L
    Gd = rng.standard_normal((ncall, cfg.B, cfg.N)).astype(np.float32)
    Gh = rng.standard_normal((cfg.B, cfg.N, cfg.D)).astype(np.float32)
    layer = ReasonGNNLayer(layer_args(cfg), batch.num_entity, cfg.num_kb_relation, cfg.D, "bfs")
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items() if not k.startswith("type_layer.")},
                          strict=True)
    layer.train()                                        # linear_dropout = 0.0 in layer_args
    inp = {k: torch.tensor(feats[k], requires_grad=True) for k in ("h0", "rel_features", "rel_features_inv", "ins")}
    layer.init_reason(local_entity=torch.from_numpy(batch.local_entity), kb_adj_mat=batch.edge_tuple,
                      local_entity_emb=inp["h0"], rel_features=inp["rel_features"],
                      rel_features_inv=inp["rel_features_inv"],
                      query_entities=torch.from_numpy(batch.query_entities).float())
    seed = torch.from_numpy(batch.seed_dist).float()
    loss = 0.0
    c = 0
    for t in range(cfg.T):
        dist = seed
        for j in range(cfg.L):
            dist, h = layer(dist, inp["ins"][t], step=j)
            # the stored forward fixtures must be what this run produced
            assert np.abs(dist.detach().numpy() - ref["dist"][c]).max() < 1e-6
            loss = loss + (dist * torch.from_numpy(Gd[c])).sum()
            c += 1
    loss = loss + (h * torch.from_numpy(Gh)).sum()
    loss.backward()
    out = {"cot.Gd": Gd, "cot.Gh": Gh, "loss": np.float64(loss.item())}
    for k, v in inp.items():
        out["grad." + k] = v.grad.numpy().copy()
    used = ("rel_linear", "e2e_linear", "score_func", "pos_emb")
    for k, p in layer.named_parameters():
        if k.startswith(used):
            out["grad." + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "grad_" + name), **out)
    print("wrote grad_" + name, "loss", loss.item(), sorted(k for k in out if k.startswith("grad.")))


def typelayer_grads():
    from modules.layer_init import TypeLayer
    z = np.load(os.path.join(HERE, "typelayer.npz"))
    B, N, D = int(z["B"]), int(z["N"]), int(z["D"])
    F = len(z["heads"])
    edge_tuple = (z["heads"], z["rels"], z["tails"], z["batch_ids"], np.arange(F, dtype=np.int64),
                  z["weight_list"].tolist(), z["weight_rel_list"].tolist())
    rng = np.random.default_rng(8)
    G = rng.standard_normal((B, N, D)).astype(np.float32)
    out = {"cot.G": G}
    for norm_rel in (False, True):
        tl = TypeLayer(D, D, torch.nn.Dropout(0.0), torch.device("cpu"), norm_rel)
        tl.load_state_dict({"kb_self_linear.weight": torch.from_numpy(z["param.type_layer.kb_self_linear.weight"]),
                            "kb_self_linear.bias": torch.from_numpy(z["param.type_layer.kb_self_linear.bias"])})
        rf = torch.tensor(z["feat.rel_features"], requires_grad=True)
        h0 = tl(local_entity=torch.from_numpy(z["local_entity"]), edge_list=edge_tuple, rel_features=rf)
        assert np.abs(h0.detach().numpy() - z["ref.h0_norm%d" % int(norm_rel)]).max() < 1e-6
        (h0 * torch.from_numpy(G)).sum().backward()
        tag = "norm%d." % int(norm_rel)
        out["grad." + tag + "rel_features"] = rf.grad.numpy().copy()
        out["grad." + tag + "kb_self_linear.weight"] = tl.kb_self_linear.weight.grad.numpy().copy()
        out["grad." + tag + "kb_self_linear.bias"] = tl.kb_self_linear.bias.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "grad_typelayer.npz"), **out)
    print("wrote grad_typelayer.npz")


if __name__ == "__main__":
    torch.manual_seed(0)
    layer_grads("layer_d200.npz")
    layer_grads("layer_d50.npz")
    typelayer_grads()
