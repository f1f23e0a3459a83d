"""ORACLE (test infrastructure, NOT product code) - library-independent numpy
restatement of the ReaRev reasoning hot path in the *factored* form the HIP
path implements ("Factored restatement"), evaluated in
float64 (or float32 on request).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.

It exists to catch "HIP path and torch restatement wrong in the same way":
it shares no code, no library call and no operation order with
``oracle/rearev_torch_cpu.py``.  Pinned, like that file, against fixtures
generated from the live reference (``tests/golden/make_golden.py``).

    T_d[r,:]     = rel_linear_step.W @ relfeat_d[r,:] + b (+ pos_emb_d_step[r,:])   reasongnn.py:75-79
    agg[n,2i+d,:]= sum_{f: dst_d(f)=n} v_f^2 * dist[src_d(f)] * relu(T_d[rel_f,:] * ins[batch(n),i,:])
                                                                                     reasongnn.py:80-84,106-111
    h'[n,:]      = relu(e2e_step.W @ concat(h[n,:], agg[n,0,:],...,agg[n,2I-1,:]) + b)  reasongnn.py:158-163
    score[n]     = score_func.w . h'[n,:] + b + (mask[n] ? 0 : -1e11)                 reasongnn.py:165-168
    dist'[g,:]   = softmax over the N slots of question g                             reasongnn.py:169
"""
from __future__ import annotations

import numpy as np

VERY_NEG_NUMBER = -100000000000.0


def segment_sum(dst: np.ndarray, vals: np.ndarray, n_rows: int) -> np.ndarray:
    """out[r] = sum of vals[f] over f with dst[f] == r, in ascending fact order
    (the canonical order of the HIP CSR)."""
    out = np.zeros((n_rows,) + vals.shape[1:], dtype=vals.dtype)
    if len(dst) == 0:
        return out
    order = np.argsort(dst, kind="stable")
    d = dst[order]
    v = vals[order]
    starts = np.flatnonzero(np.r_[True, d[1:] != d[:-1]])
    out[d[starts]] = np.add.reduceat(v, starts, axis=0)
    return out


def edge_weight(edge_tuple, normalized: bool, dtype):
    heads = np.asarray(edge_tuple[0])
    if normalized:
        return np.asarray(edge_tuple[5], dtype=np.float32).astype(dtype)
    return np.ones(len(heads), dtype=dtype)


def layer_call(edge_tuple, B, N, h, mask, dist, ins, params, step, relfeat, relfeat_inv,
               normalized_gnn=False, use_posemb=False, dtype=np.float64):
    heads = np.asarray(edge_tuple[0], dtype=np.int64)
    rels = np.asarray(edge_tuple[1], dtype=np.int64)
    tails = np.asarray(edge_tuple[2], dtype=np.int64)
    D = h.shape[-1]
    I = ins.shape[1]
    v = edge_weight(edge_tuple, normalized_gnn, dtype)
    W_r = params[f"rel_linear{step}.weight"].astype(dtype)
    b_r = params[f"rel_linear{step}.bias"].astype(dtype)
    W_e = params[f"e2e_linear{step}.weight"].astype(dtype)
    b_e = params[f"e2e_linear{step}.bias"].astype(dtype)
    p = dist.reshape(-1).astype(dtype)
    hh = h.reshape(B * N, D).astype(dtype)
    q = ins.astype(dtype)
    agg = np.zeros((B * N, 2 * I, D), dtype=dtype)
    for d, (feat, src, dst, pkey) in enumerate((
            (relfeat, heads, tails, f"pos_emb{step}.weight"),
            (relfeat_inv, tails, heads, f"pos_emb_inv{step}.weight"))):
        T = feat.astype(dtype) @ W_r.T + b_r
        if use_posemb:
            pe = params[pkey].astype(dtype)
            T[: pe.shape[0]] += pe        # pos_emb has num_kb_relation rows; the pad row is never indexed
        coef = (v * v) * p[src]           # weight enters both sparse products (base_gnn.py:38-47)
        nz = np.flatnonzero(coef != 0)    # facts with zero prior contribute exact zeros
        g = dst[nz] // N
        for i in range(I):
            msg = np.maximum(T[rels[nz]] * q[g, i, :], 0) * coef[nz, None]
            agg[:, 2 * i + d, :] = segment_sum(dst[nz], msg, B * N)
    x = np.concatenate([hh, agg.reshape(B * N, 2 * I * D)], axis=1)
    h_new = np.maximum(x @ W_e.T + b_e, 0)
    score = h_new @ params["score_func.weight"].astype(dtype)[0] + params["score_func.bias"].astype(dtype)[0]
    # The mask is added in float32 in the reference: |score| << ulp(1e11) = 8192, so a masked
    # slot becomes exactly -1e11 and an all-masked question softmaxes to exactly uniform 1/N.
    # That rounding is part of the observable behaviour, so it is reproduced here.
    m32 = (1 - mask.astype(np.float32)) * np.float32(VERY_NEG_NUMBER)
    score = (score.reshape(B, N).astype(np.float32) + m32).astype(dtype)
    m = score.max(axis=1, keepdims=True)
    e = np.exp(score - m)
    new_dist = e / e.sum(axis=1, keepdims=True)
    return score, new_dist, h_new.reshape(B, N, D), agg


def type_layer(edge_tuple, B, N, rel_features, W, b, norm_rel, dtype=np.float64):
    """h0 = relu( sum_{tail=n} v T[rel] + sum_{head=n} v T[rel] ), T = W relfeat + b
    (layer_init.py:39-57), v = weight_rel_list if norm_rel else 1."""
    heads = np.asarray(edge_tuple[0], dtype=np.int64)
    rels = np.asarray(edge_tuple[1], dtype=np.int64)
    tails = np.asarray(edge_tuple[2], dtype=np.int64)
    if norm_rel:
        v = np.asarray(edge_tuple[6], dtype=np.float32).astype(dtype)
    else:
        v = np.ones(len(heads), dtype=dtype)
    T = rel_features.astype(dtype) @ W.astype(dtype).T + b.astype(dtype)
    msg = T[rels] * v[:, None]
    out = segment_sum(tails, msg, B * N) + segment_sum(heads, msg, B * N)
    return np.maximum(out, 0).reshape(B, N, -1)


def run_stack(batch, feats, params, *, use_type_layer=False, norm_rel=False, dtype=np.float64):
    """Same driving loop as rearev.py:206-211 (see rearev_torch_cpu.run_stack)."""
    cfg = batch.cfg
    mask = (batch.local_entity != batch.num_entity).astype(dtype)
    seed = batch.seed_dist.astype(np.float32).astype(dtype)
    out = {"score": [], "dist": [], "h": [], "agg": []}
    if use_type_layer:
        h = type_layer(batch.edge_tuple, cfg.B, cfg.N, feats["rel_features"],
                       params["type_layer.kb_self_linear.weight"],
                       params["type_layer.kb_self_linear.bias"], norm_rel, dtype)
        out["h0"] = h
    else:
        h = feats["h0"].astype(dtype)
    for t in range(cfg.T):
        dist = seed
        for j in range(cfg.L):
            score, dist, h, agg = layer_call(
                batch.edge_tuple, cfg.B, cfg.N, h, mask, dist, feats["ins"][t], params, j,
                feats["rel_features"], feats["rel_features_inv"],
                normalized_gnn=cfg.normalized_gnn, use_posemb=cfg.pos_emb, dtype=dtype)
            out["score"].append(score)
            out["dist"].append(dist)
            out["h"].append(h)
            out["agg"].append(agg)
    return out
