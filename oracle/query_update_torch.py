"""ORACLE (test infrastructure, NOT product code) - op-for-op torch restatement of
``QueryReform.forward`` / ``Fusion.forward`` (reference ``gnn/modules/query_update.py:13-16,26-44``),
INCLUDING the attention over all nodes that the reference computes and then discards (``:36-38``).
Device agnostic: tests run it on the GPU next to the drop-in to check that dropping the dead code
changes nothing, and to time what the reference's op sequence costs there.
Only ``tests/`` may import this file."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def fusion(x, y, W_r, W_g):
    cat = torch.cat([x, y, x - y], dim=-1)                                   # query_update.py:14-15
    r_ = F.linear(cat, W_r)
    g_ = torch.sigmoid(F.linear(cat, W_g))
    return g_ * r_ + (1 - g_) * x                                            # :16


def query_reform(q_node, ent_emb, seed_info, ent_mask, W_attn, b_attn, W_r, W_g):
    q_ent_attn = (F.linear(q_node, W_attn, b_attn).unsqueeze(1) * ent_emb).sum(2, keepdim=True)    # :36
    q_ent_attn = F.softmax(q_ent_attn - (1 - ent_mask.unsqueeze(2)) * 1e8, dim=1)                 # :37
    attn_retrieve = (q_ent_attn * ent_emb).sum(1)                                                  # :38 (unused)
    seed_retrieve = torch.bmm(seed_info.unsqueeze(1), ent_emb).squeeze(1)                          # :40
    del attn_retrieve
    return fusion(q_node, seed_retrieve, W_r, W_g)                                                 # :44
