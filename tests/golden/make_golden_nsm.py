#!/usr/bin/env python
"""Generates tests/golden/nsm_layer.npz from the LIVE reference's ``NSMLayer``
(``gnn/modules/kg_reasoning/nsm_gnn.py``): outputs of ``num_step`` chained layer calls and the gradients
autograd derives through them, for reason_kb / normalized_gnn off and on.

    python tests/golden/make_golden_nsm.py          (build container only, CPU)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/gnn")

import gnnrag_amd  # noqa: E402,F401
from gnnrag_amd import synth  # noqa: E402


def main():
    from modules.kg_reasoning.nsm_gnn import NSMLayer, NSMLayer_back
    cfg = synth.GraphConfig(name="nsm", B=4, N=44, E=130, R=8, D=64, I=1, L=3, seed=31, n_real_min=6)
    batch = synth.make_batch(cfg)
    rng = np.random.default_rng(31)
    D, L = cfg.D, cfg.L
    h0 = (0.1 * rng.standard_normal((cfg.B, cfg.N, D))).astype(np.float32)
    relfeat = (0.3 * rng.standard_normal((cfg.R1, D))).astype(np.float32)
    ins = (0.3 * rng.standard_normal((L, cfg.B, 1, D))).astype(np.float32)
    Gd = rng.standard_normal((L, cfg.B, cfg.N)).astype(np.float32)
    Gh = rng.standard_normal((cfg.B, cfg.N, D)).astype(np.float32)
    out = dict(B=cfg.B, N=cfg.N, D=D, L=L, R1=cfg.R1, num_entity=batch.num_entity, local_entity=batch.local_entity,
               seed_dist=batch.seed_dist, heads=np.asarray(batch.edge_tuple[0]), rels=np.asarray(batch.edge_tuple[1]),
               tails=np.asarray(batch.edge_tuple[2]), weight_list=np.asarray(batch.edge_tuple[5], np.float64),
               h0=h0, rel_features=relfeat, ins=ins, Gd=Gd, Gh=Gh)
    torch.manual_seed(31)
    proto = None
    relfeat_inv = (0.3 * rng.standard_normal((cfg.R1, D))).astype(np.float32)
    out["rel_features_inv"] = relfeat_inv
    for tag, reason_kb, normalized, cls in (("plain", False, False, NSMLayer), ("kb_norm", True, True, NSMLayer),
                                            ("back_plain", False, False, NSMLayer_back),
                                            ("back_kb_norm", True, True, NSMLayer_back)):
        back = cls is NSMLayer_back
        args = dict(use_cuda=False, normalized_gnn=normalized, num_step=L, reason_kb=reason_kb, linear_dropout=0.0)
        layer = cls(args, batch.num_entity, cfg.num_kb_relation, D)
        if proto is None:
            proto = {k: v.detach().clone() for k, v in layer.state_dict().items()}
            for k, v in proto.items():
                out["param." + k] = v.numpy()
        layer.load_state_dict(proto)
        layer.train()
        X = {"h0": torch.tensor(h0, requires_grad=True),
             "rel_features": torch.tensor(relfeat_inv if back else relfeat, requires_grad=True),
             "ins": torch.tensor(ins, requires_grad=True)}
        layer.init_reason(local_entity=torch.from_numpy(batch.local_entity), kb_adj_mat=batch.edge_tuple,
                          local_entity_emb=X["h0"], rel_features=X["rel_features"])
        if back:
            # NSMLayer_back.reason_layer reads self.rel_features_inv (nsm_gnn.py:122), which the reference's own
            # init_reason never sets; its caller (models/NSM/nsm.py) has to assign it, and so does this script
            layer.rel_features_inv = X["rel_features"]
        dist = torch.from_numpy(batch.seed_dist).float()
        loss = 0.0
        rec = {"score": [], "dist": [], "h": []}
        for j in range(L):
            score, dist = layer(dist, X["ins"][j], step=j, return_score=True)
            rec["score"].append(score.detach().numpy().copy())
            rec["dist"].append(dist.detach().numpy().copy())
            rec["h"].append(layer.local_entity_emb.detach().numpy().copy())
            loss = loss + (dist * torch.from_numpy(Gd[j])).sum()
        loss = loss + (layer.local_entity_emb * torch.from_numpy(Gh)).sum()
        loss.backward()
        for k, lst in rec.items():
            out["%s.ref.%s" % (tag, k)] = np.stack(lst)
        out["%s.loss" % tag] = np.float64(loss.item())
        for k, v in X.items():
            out["%s.grad.%s" % (tag, k)] = v.grad.numpy().copy()
        for k, p in layer.named_parameters():
            if p.grad is not None:
                out["%s.grad.%s" % (tag, k)] = p.grad.numpy().copy()
        out["%s.possible_cand" % tag] = np.stack([m.numpy() for m in layer.possible_cand])
    np.savez_compressed(os.path.join(HERE, "nsm_layer.npz"), **out)
    print("wrote nsm_layer.npz", sorted(k for k in out if ".grad." in k)[:8], "...")


if __name__ == "__main__":
    main()
