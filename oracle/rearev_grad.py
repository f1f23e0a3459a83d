"""ORACLE (test infrastructure, NOT product code) - float64 autograd restatement of the
typed-edge aggregation and of TypeLayer's sparse part, for checking the HIP backward kernels
(``gnnrag_aggregate_backward`` / ``gnnrag_typelayer_backward``).

Only ``tests/`` may import this file.  The product (``gnn-rag_amd/``) never does.

The forward below is the factored form of ``reason_layer`` / ``reason_layer_inv`` (reference
``gnn/modules/kg_reasoning/reasongnn.py:61-116``, concat order of ``:150-158``) and of
``TypeLayer.forward`` (``gnn/modules/layer_init.py:47-57``) written with differentiable torch ops
in float64; the gradients are whatever ``torch.autograd`` derives from it - the same engine the
reference trains with (``train_model.py:222-228``).

Pinning: the end-to-end gradients of the *live reference modules* are recorded by
``tests/golden/make_golden_grad.py`` (fixtures ``tests/golden/grad_*.npz``); this restatement is
checked against those on CPU (``tests/test_oracle_golden.py``) before the GPU tests rely on it.
"""
from __future__ import annotations

import numpy as np
import torch

F64 = torch.float64


def _idx(edge_tuple, N):
    heads = torch.as_tensor(np.asarray(edge_tuple[0], np.int64))
    rels = torch.as_tensor(np.asarray(edge_tuple[1], np.int64))
    tails = torch.as_tensor(np.asarray(edge_tuple[2], np.int64))
    return heads, rels, tails, heads // N


def aggregate(edge_tuple, B, N, dist, ins, T_fwd, T_inv, weight=None):
    """agg [B*N, 2I*D]; dist [B,N], ins [B,I,D], T_* [R1,D] float64 tensors (may require grad).
    weight: per-fact v_f (the reference applies it in both sparse products -> v_f^2,
    base_gnn.py:38-47) or None."""
    heads, rels, tails, bid = _idx(edge_tuple, N)
    I, D = ins.shape[1], ins.shape[2]
    w = torch.ones(len(heads), dtype=F64) if weight is None else torch.as_tensor(np.asarray(weight), dtype=F64) ** 2
    flat = dist.reshape(-1)
    blocks = []
    for i in range(I):
        for src, dst, T in ((heads, tails, T_fwd), (tails, heads, T_inv)):
            msg = torch.relu(T[rels] * ins[bid, i])                       # reasongnn.py:71-79 / :98-105
            val = (w * flat[src]).unsqueeze(1) * msg                      # :80-82 / :106-108
            blocks.append(torch.zeros(B * N, D, dtype=F64).index_add(0, dst, val))   # :84 / :111
    return torch.cat(blocks, dim=1)                                       # :150-158


def aggregate_grads(edge_tuple, B, N, dist, ins, T_fwd, T_inv, g_agg, weight=None):
    """numpy in, numpy out: (agg, g_dist [B*N], g_ins, g_T_fwd, g_T_inv) for the cotangent g_agg."""
    with torch.enable_grad():            # callable from inside another backward pass (grad mode is off there)
        t = [torch.tensor(np.asarray(x, np.float64), requires_grad=True) for x in (dist, ins, T_fwd, T_inv)]
        agg = aggregate(edge_tuple, B, N, t[0].reshape(B, N), t[1], t[2], t[3], weight)
        agg.backward(torch.as_tensor(np.asarray(g_agg, np.float64)).reshape(agg.shape))
    return (agg.detach().numpy(), t[0].grad.reshape(-1).numpy(), t[1].grad.numpy(), t[2].grad.numpy(),
            t[3].grad.numpy())


def typelayer_pre(edge_tuple, B, N, T, weight_rel=None):
    """pre-activation of TypeLayer: sum over incident facts (tail side + head side) of v_f T[rel_f]."""
    heads, rels, tails, _ = _idx(edge_tuple, N)
    v = torch.ones(len(heads), dtype=F64) if weight_rel is None else torch.as_tensor(np.asarray(weight_rel), dtype=F64)
    val = v.unsqueeze(1) * T[rels]                                        # layer_init.py:47-49
    z = torch.zeros(B * N, T.shape[1], dtype=F64)
    return z.index_add(0, tails, val) + z.index_add(0, heads, val)        # layer_init.py:53-57


def typelayer_grad(edge_tuple, B, N, T, g_pre, weight_rel=None):
    with torch.enable_grad():
        t = torch.tensor(np.asarray(T, np.float64), requires_grad=True)
        typelayer_pre(edge_tuple, B, N, t, weight_rel).backward(torch.as_tensor(np.asarray(g_pre, np.float64)))
    return t.grad.numpy()


VERY_NEG_NUMBER = -100000000000  # reasongnn.py:9


def stack_grads(batch, feats, params, Gd, Gh):
    """Gradients of  sum_c <dist_c, Gd[c]> + <h_last, Gh>  over the T*L layer calls of the layer
    stack (reasongnn.py:134-174 chained as rearev.py:206-211 does), float64.  Returns
    {name: grad} with the reference's parameter names plus the inputs h0 / rel_features /
    rel_features_inv / ins."""
    cfg = batch.cfg
    B, N, D, I, L, T = cfg.B, cfg.N, cfg.D, cfg.I, cfg.L, cfg.T
    used = ("rel_linear", "e2e_linear", "score_func", "pos_emb")
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in params.items()
         if k.startswith(used)}
    X = {k: torch.tensor(np.asarray(feats[k], np.float64), requires_grad=True)
         for k in ("h0", "rel_features", "rel_features_inv", "ins")}
    mask32 = torch.as_tensor((batch.local_entity != batch.num_entity).astype(np.float32))
    seed = torch.as_tensor(batch.seed_dist.astype(np.float64))
    weight = batch.edge_tuple[5] if cfg.normalized_gnn else None
    h = X["h0"]
    loss = 0.0
    c = 0
    for t in range(T):
        dist = seed
        for j in range(L):
            Wr, br = P["rel_linear%d.weight" % j], P["rel_linear%d.bias" % j]
            Tf = X["rel_features"] @ Wr.T + br
            Ti = X["rel_features_inv"] @ Wr.T + br
            if cfg.pos_emb:
                pos, posi = P["pos_emb%d.weight" % j], P["pos_emb_inv%d.weight" % j]
                pad = torch.zeros(Tf.shape[0] - pos.shape[0], D, dtype=F64)
                Tf = Tf + torch.cat([pos, pad])
                Ti = Ti + torch.cat([posi, pad])
            agg = aggregate(batch.edge_tuple, B, N, dist, X["ins"][t], Tf, Ti, weight)
            x = torch.cat([h.reshape(B * N, D), agg], dim=1)
            h = torch.relu(x @ P["e2e_linear%d.weight" % j].T + P["e2e_linear%d.bias" % j]).reshape(B, N, D)
            score = h @ P["score_func.weight"].reshape(-1) + P["score_func.bias"]
            # the reference adds the mask term in fp32 (reasongnn.py:166-168): a masked score becomes
            # exactly -1e11 in the forward while autograd still passes the gradient straight through
            score = (score.to(torch.float32) + (1 - mask32) * VERY_NEG_NUMBER).to(F64)
            dist = torch.softmax(score, dim=1)
            loss = loss + (dist * torch.as_tensor(np.asarray(Gd[c], np.float64))).sum()
            c += 1
    loss = loss + (h * torch.as_tensor(np.asarray(Gh, np.float64))).sum()
    loss.backward()
    out = {k: v.grad.numpy() for k, v in P.items()}
    out.update({k: v.grad.numpy() for k, v in X.items()})
    out["loss"] = float(loss.item())
    return out
