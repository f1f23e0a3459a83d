"""Drop-in for the reference's ``modules/query_update.py``: the instruction
update between two ReaRev iterations (``rearev.py:217-221``).

``QueryReform.forward`` in the reference (``query_update.py:26-44``) first computes an attention over
all N node states (``:36-38``: a [B,N,D] product, a softmax over N and a second [B,N,D] product - three
passes over the node state plus two [B,N,D] temporaries) and then does not use it: the value it returns
is ``fusion(q_node, seed_retrieve)`` (``:40,44``).  Here only that is computed, and ``seed_retrieve`` reads
just the seed rows instead of streaming the node state through a bmm - retrieval and Fusion in ONE launch
(``gnnrag_query_reform``; ``gnnrag_seed_retrieve`` + the torch Fusion for shapes it does not take).  Same
classes, constructors, parameter names (``q_ent_attn`` is kept: released checkpoints hold it) and
return values.  With autograd enabled the seed retrieval is the reference's ``torch.bmm``."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class Fusion(nn.Module):
    """Gated mix of an instruction x with retrieved evidence y (reference: query_update.py:6-16)."""

    def __init__(self, d_hid):
        super().__init__()
        self.r = nn.Linear(d_hid * 3, d_hid, bias=False)
        self.g = nn.Linear(d_hid * 3, d_hid, bias=False)

    def forward(self, x, y):
        feats = torch.cat([x, y, x - y], dim=-1)
        gate = torch.sigmoid(self.g(feats))
        return gate * self.r(feats) + (1 - gate) * x


class QueryReform(nn.Module):
    """Instruction update from the seeds' node states (reference: query_update.py:18-44)."""

    def __init__(self, h_dim):
        super().__init__()
        self.fusion = Fusion(h_dim)
        self.q_ent_attn = nn.Linear(h_dim, h_dim)      # unused by the returned value; state_dict parity

    def forward(self, q_node, ent_emb, seed_info, ent_mask):
        if torch.is_grad_enabled():
            seed_retrieve = torch.bmm(seed_info.unsqueeze(1), ent_emb).squeeze(1)       # :40 (autograd form)
        else:
            base = getattr(ent_emb, "_gnnrag_padded", None)        # node state kept zero-padded by ReasonGNNLayer
            D = ent_emb.shape[-1]
            r, g = self.fusion.r, self.fusion.g
            if (q_node.dim() == 2 and q_node.shape[-1] == D and D <= 4096 and r.bias is None and g.bias is None
                    and r.weight.dtype == torch.float32 and g.weight.dtype == torch.float32):
                # retrieval + Fusion (:40,44 with :6-16) in one launch: ~11 small torch launches per call otherwise
                return ops.query_reform(q_node.detach().float(), seed_info.float(), base if base is not None else ent_emb.float(),
                                        r.weight.detach(), g.weight.detach())
            if base is not None:
                seed_retrieve = ops.seed_retrieve(seed_info.float(), base)[:, : ent_emb.shape[-1]]
            else:
                seed_retrieve = ops.seed_retrieve(seed_info.float(), ent_emb.float())   # raises on CPU tensors
        return self.fusion(q_node, seed_retrieve)                                       # :44


class AttnEncoder(nn.Module):
    """Masked attention pooling over a sequence (reference: query_update.py:46-62)."""

    def __init__(self, d_hid):
        super().__init__()
        self.attn_linear = nn.Linear(d_hid, 1, bias=False)

    def forward(self, x, x_mask):
        logits = self.attn_linear(x) - (1 - x_mask.unsqueeze(2)) * 1e8
        return (x * F.softmax(logits, dim=1)).sum(1)
