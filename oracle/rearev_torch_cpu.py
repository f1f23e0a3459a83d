"""ORACLE (test infrastructure, NOT product code) - CPU restatement of the
GNN-RAG ReaRev reasoning hot path, op for op, on PyTorch-CPU.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product (``gnn-rag_amd/``) never does.

Why torch: the reference's arithmetic for this path *is* stock PyTorch
(``torch.sparse.mm``, ``index_select``, ``addmm``, ``softmax``; reference pins
torch==1.7.1, ``gnn/requirements.txt:3``), i.e. a third-party dependency that is
not under ``/root/reference``.  Restating the path with the same library calls
keeps the CPU op mix - and therefore the ``cpu_baseline`` timing - that of the
reference.  ``oracle/rearev_np64.py`` is the second, library-independent
restatement (factored form, float64).

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4).  This restatement is pinned against outputs of the *live
reference modules* imported from ``/root/reference/gnn`` (in the build
container) by ``tests/golden/make_golden.py``; the resulting fixtures are
committed under ``tests/golden/*.npz`` and checked by
``tests/test_oracle_golden.py``.

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import numpy as np
import torch

VERY_NEG_NUMBER = -100000000000  # reasongnn.py:9


def _t(x, dtype):
    return torch.as_tensor(np.asarray(x), dtype=dtype)


class Structure:
    """The sparse operators of one batch - follows ``BaseGNNLayer.build_matrix``
    (``gnn/modules/kg_reasoning/base_gnn.py:19-51``).  Only the four matrices
    ReaRev uses are built (fact->tail, head->fact, fact->head, tail->fact)."""

    def __init__(self, edge_tuple, B: int, N: int, normalized_gnn: bool):
        heads, rels, tails, bids, fact_ids, weight_list, _ = edge_tuple
        F = len(fact_ids)
        self.B, self.N, self.F = B, N, F
        h = _t(heads, torch.long)
        t = _t(tails, torch.long)
        f = _t(fact_ids, torch.long)
        self.rels = _t(rels, torch.long)          # base_gnn.py:33
        self.bids = _t(bids, torch.long)          # base_gnn.py:34
        if normalized_gnn:                        # base_gnn.py:38-41
            v = torch.tensor(list(weight_list), dtype=torch.float32)
        else:
            v = torch.ones(F, dtype=torch.float32)
        nn_ = B * N

        def coo(r, c, shape):
            return torch.sparse_coo_tensor(torch.stack([r, c]), v, shape)

        self.fact2head = coo(h, f, (nn_, F))      # base_gnn.py:44
        self.head2fact = coo(f, h, (F, nn_))      # base_gnn.py:45
        self.fact2tail = coo(t, f, (nn_, F))      # base_gnn.py:46
        self.tail2fact = coo(f, t, (F, nn_))      # base_gnn.py:47


def _direction(st: Structure, relfeat, dist, ins_i, W, b, pos, src2fact, fact2dst):
    """One directed, one-instruction aggregation.  Follows
    ``ReasonGNNLayer.reason_layer`` / ``reason_layer_inv``
    (``reasongnn.py:61-89`` / ``:91-116``)."""
    fact_rel = relfeat.index_select(0, st.rels)                 # :71 / :98
    fact_query = ins_i.index_select(0, st.bids)                 # :73 / :100
    lin = torch.nn.functional.linear(fact_rel, W, b)            # rel_linear(...)
    if pos is not None:
        lin = lin + pos.index_select(0, st.rels)                # :75-77
    fact_val = torch.relu(lin * fact_query)                     # :77/:79
    fact_prior = torch.sparse.mm(src2fact, dist.reshape(-1, 1))  # :80 / :106
    fact_val = fact_val * fact_prior                            # :82 / :109
    out = torch.sparse.mm(fact2dst, fact_val)                   # :84 / :111
    return out.view(st.B, st.N, -1)                             # :87 / :114


def layer_forward(st: Structure, h, mask, dist, ins, params, step: int,
                  relfeat, relfeat_inv, use_posemb: bool):
    """One ``ReasonGNNLayer.forward`` call (``reasongnn.py:134-174``), eval mode
    (dropout = identity).  Returns (score_tp, dist', h')."""
    W_r = params[f"rel_linear{step}.weight"]
    b_r = params[f"rel_linear{step}.bias"]
    W_e = params[f"e2e_linear{step}.weight"]
    b_e = params[f"e2e_linear{step}.bias"]
    pos = params[f"pos_emb{step}.weight"] if use_posemb else None
    pos_inv = params[f"pos_emb_inv{step}.weight"] if use_posemb else None
    reps = []
    for i in range(ins.shape[1]):                                # :150-156
        reps.append(_direction(st, relfeat, dist, ins[:, i, :], W_r, b_r, pos,
                               st.head2fact, st.fact2tail))
        reps.append(_direction(st, relfeat_inv, dist, ins[:, i, :], W_r, b_r, pos_inv,
                               st.tail2fact, st.fact2head))
    x = torch.cat([h] + reps, dim=2)                             # :158-161
    h_new = torch.relu(torch.nn.functional.linear(x, W_e, b_e))  # :163
    score = torch.nn.functional.linear(
        h_new, params["score_func.weight"], params["score_func.bias"]).squeeze(2)  # :165
    score = score + (1 - mask) * VERY_NEG_NUMBER                 # :168
    new_dist = torch.softmax(score, dim=1)                       # :169
    return score, new_dist, h_new


def type_layer(edge_tuple, B: int, N: int, rel_features, W, b, norm_rel: bool):
    """``TypeLayer.forward`` (``gnn/modules/layer_init.py:25-62``)."""
    heads, rels, tails, bids, fact_ids, _, weight_rel_list = edge_tuple
    F = len(fact_ids)
    h = _t(heads, torch.long)
    t = _t(tails, torch.long)
    f = _t(fact_ids, torch.long)
    r = _t(rels, torch.long)
    if norm_rel:                                                 # :39-42
        v = torch.tensor(list(weight_rel_list), dtype=torch.float32)
    else:
        v = torch.ones(F, dtype=torch.float32)
    fact_val = torch.nn.functional.linear(rel_features.index_select(0, r), W, b)  # :47-49
    f2t = torch.sparse_coo_tensor(torch.stack([t, f]), v, (B * N, F))             # :53
    f2h = torch.sparse_coo_tensor(torch.stack([h, f]), v, (B * N, F))             # :54
    out = torch.relu(torch.sparse.mm(f2t, fact_val) + torch.sparse.mm(f2h, fact_val))  # :57
    return out.view(B, N, -1)


def to_torch_params(params: dict) -> dict:
    return {k: _t(v, torch.float32) for k, v in params.items()}


def run_stack(batch, feats: dict, params: dict, *, use_type_layer: bool = False,
              norm_rel: bool = False, n_threads: int | None = None) -> dict:
    """Runs T outer iterations x L layer calls the way ``ReaRev.forward`` drives the
    layer (``gnn/models/ReaRev/rearev.py:206-211``): ``dist`` is reset to the seed
    distribution at the top of every iteration, node embeddings carry over.
    ``feats['ins'][t]`` stands in for the instructions of iteration t."""
    cfg = batch.cfg
    if n_threads:
        torch.set_num_threads(n_threads)
    p = to_torch_params(params)
    relfeat = _t(feats["rel_features"], torch.float32)
    relfeat_inv = _t(feats["rel_features_inv"], torch.float32)
    ins_all = _t(feats["ins"], torch.float32)
    local_entity = _t(batch.local_entity, torch.long)
    mask = (local_entity != batch.num_entity).float()            # reasongnn.py:48
    seed = _t(batch.seed_dist, torch.float32)                    # rearev.py:174
    out = {"score": [], "dist": [], "h": []}
    with torch.no_grad():
        if use_type_layer:
            h = type_layer(batch.edge_tuple, cfg.B, cfg.N, relfeat,
                           p["type_layer.kb_self_linear.weight"],
                           p["type_layer.kb_self_linear.bias"], norm_rel)
            out["h0"] = h.numpy().copy()
        else:
            h = _t(feats["h0"], torch.float32)
        st = Structure(batch.edge_tuple, cfg.B, cfg.N, cfg.normalized_gnn)
        for t in range(cfg.T):
            dist = seed                                           # rearev.py:208
            for j in range(cfg.L):                                # rearev.py:209-210
                score, dist, h = layer_forward(st, h, mask, dist, ins_all[t], p, j,
                                               relfeat, relfeat_inv, cfg.pos_emb)
                out["score"].append(score.numpy().copy())
                out["dist"].append(dist.numpy().copy())
                out["h"].append(h.numpy().copy())
    return out
