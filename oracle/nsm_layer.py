"""ORACLE (test infrastructure, NOT product code) - float64 restatement of the reference's NSM layer
(``gnn/modules/kg_reasoning/nsm_gnn.py:53-78`` forward, ``:87-112`` NSMLayer.reason_layer,
``:118-142`` NSMLayer_back.reason_layer) on top of ``oracle/rearev_grad.aggregate`` (differentiable).
Only ``tests/`` may import this file.  Pinned against ``tests/golden/nsm_layer.npz`` (outputs and
gradients of the live reference's NSMLayer, ``tests/golden/make_golden_nsm.py``)."""
from __future__ import annotations

import numpy as np
import torch

from . import rearev_grad as og

F64 = torch.float64
VERY_SMALL_NUMBER = 1e-10
VERY_NEG_NUMBER = -100000000000


def run(edge_tuple, B, N, local_entity, num_entity, h0, rel_features, ins_steps, seed_dist, params, *,
        reason_kb, normalized_gnn, direction=0, Gd=None, Gh=None):
    """L = len(ins_steps) chained layer calls (nsm.py:219-221).  Returns dict(score, dist, h) lists and, if
    cotangents are given, gradients of  sum_c <dist_c, Gd[c]> + <h_last, Gh>."""
    D = h0.shape[-1]
    need_grad = Gd is not None
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=need_grad) for k, v in params.items()}
    X = {"h0": torch.tensor(np.asarray(h0, np.float64), requires_grad=need_grad),
         "rel_features": torch.tensor(np.asarray(rel_features, np.float64), requires_grad=need_grad),
         "ins": torch.tensor(np.asarray(ins_steps, np.float64), requires_grad=need_grad)}
    mask32 = torch.as_tensor((np.asarray(local_entity) != num_entity).astype(np.float32))
    heads = torch.as_tensor(np.asarray(edge_tuple[0], np.int64))
    tails = torch.as_tensor(np.asarray(edge_tuple[2], np.int64))
    src, dst = (heads, tails) if direction == 0 else (tails, heads)
    weight = edge_tuple[5] if normalized_gnn else None
    w = torch.ones(len(heads), dtype=F64) if weight is None else torch.as_tensor(np.asarray(weight), dtype=F64) ** 2
    h = X["h0"]
    dist = torch.as_tensor(np.asarray(seed_dist, np.float64))
    out = {"score": [], "dist": [], "h": []}
    loss = 0.0
    for j in range(len(ins_steps)):
        T = X["rel_features"] @ P["rel_linear%d.weight" % j].T + P["rel_linear%d.bias" % j]
        agg = og.aggregate(edge_tuple, B, N, dist, X["ins"][j].reshape(B, 1, D), T, T, weight)
        nbr = agg[:, direction * D:(direction + 1) * D]                            # nsm_gnn.py:108 / :136
        reach = torch.zeros(B * N, dtype=F64).index_add(0, dst, w * dist.reshape(-1)[src].detach())
        possible = (reach > VERY_SMALL_NUMBER).float().reshape(B, N)               # :101-105
        x = torch.cat([h.reshape(B * N, D), nbr], dim=1)
        h = torch.relu(x @ P["e2e_linear%d.weight" % j].T + P["e2e_linear%d.bias" % j]).reshape(B, N, D)   # :63-66
        score = h @ P["score_func.weight"].reshape(-1) + P["score_func.bias"]
        answer_mask = mask32 * possible if reason_kb else mask32                   # :69-72
        score = (score.to(torch.float32) + (1 - answer_mask) * VERY_NEG_NUMBER).to(F64)    # fp32 add, as the reference
        dist = torch.softmax(score, dim=1)
        out["score"].append(score.detach().numpy())
        out["dist"].append(dist.detach().numpy())
        out["h"].append(h.detach().numpy())
        if need_grad:
            loss = loss + (dist * torch.as_tensor(np.asarray(Gd[j], np.float64))).sum()
    if need_grad:
        loss = loss + (h * torch.as_tensor(np.asarray(Gh, np.float64))).sum()
        loss.backward()
        out["grad"] = {k: v.grad.numpy() for k, v in P.items() if v.grad is not None}
        out["grad"].update({k: v.grad.numpy() for k, v in X.items()})
        out["loss"] = float(loss.item())
    return out
