"""Drop-in for the reference's ``modules/layer_init.py`` (``TypeLayer``).

h0 = relu( A_tail (W r + b) + A_head (W r + b) )  - reference ``layer_init.py:25-62``.
W r + b is computed once per relation row on fp32 MFMA (``gnnrag_linear``), the two
sparse products become one CSR walk (``gnnrag_typelayer``).  With autograd enabled the walk
is an autograd function whose backward is ``gnnrag_typelayer_backward``."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..autograd import TypeAggFn
from .kg_reasoning import base_gnn
from .kg_reasoning.base_gnn import plan_for

VERY_NEG_NUMBER = -100000000000
VERY_SMALL_NUMBER = 1e-10


class TypeLayer(nn.Module):
    """Initial entity embeddings from incident relation types (reference: layer_init.py:9-65)."""

    def __init__(self, in_features, out_features, linear_drop, device, norm_rel):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.linear_drop = linear_drop
        self.kb_self_linear = nn.Linear(in_features, out_features)
        self.device = device
        self.norm_rel = norm_rel

    def forward(self, local_entity, edge_list, rel_features):
        base_gnn._check_gpu_tensor(rel_features, "gnnrag_amd.TypeLayer")
        batch_size, max_local_entity = local_entity.size()
        plan = plan_for(edge_list, batch_size, max_local_entity, rel_features.size(0), rel_features.device)
        if self.norm_rel:
            plan.attach_w_rel(edge_list[6])
        if torch.is_grad_enabled():
            # training: kb_self_linear is an nn.Linear call autograd knows; the sparse part and its
            # backward are HIP (gnnrag_typelayer / gnnrag_typelayer_backward)
            h0 = TypeAggFn.apply(plan, self.kb_self_linear(rel_features.float()), bool(self.norm_rel))
        else:
            T = ops.linear(rel_features.detach().float(), self.kb_self_linear.weight, self.kb_self_linear.bias)
            h0 = ops.typelayer(plan, T, bool(self.norm_rel))
        return h0.view(batch_size, max_local_entity, self.out_features)
