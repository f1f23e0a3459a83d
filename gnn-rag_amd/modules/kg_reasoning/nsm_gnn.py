"""Drop-in for the reference's ``modules/kg_reasoning/nsm_gnn.py`` (the NSM
layer is the ReaRev layer with ONE instruction and ONE direction, plus an optional reachability mask).

    neighbor[n,:]   = sum_{f: tail_f=n} w_f dist[head_f] relu(rel_linear(rel_features[rel_f]) * ins[b,:])
    possible_tail[n] = (sum_{f: tail_f=n} w_f dist[head_f]) > 1e-10
    h'              = relu(e2e_linear([h | neighbor]));  score = score_func(h') + (1 - mask) * -1e11
    mask            = local_entity_mask (* possible_tail if reason_kb)          (nsm_gnn.py:53-78,87-112)

No new kernel: with ``W' = [W_self | W_nbr | 0]`` (``[W_self | 0 | W_nbr]`` for the backward layer) the fused
ReaRev path (``gnnrag_reason_layer``, I = 1) computes exactly this - the other direction's relation
tables are ``0 . relu(..)`` = 0 - and ``possible_tail`` is the same walk over a table of ones.  The call carries
``GNNRAG_PATH_ONLY_FWD`` / ``_INV``: where the V-form table kernel and the LDS walk apply, the unused direction's
tables are neither built nor walked (bit-identical results, tests/test_gpu_parity.py).  Same
class names, constructors, parameter names and return values as the reference.  With autograd enabled
the aggregation is the HIP autograd function, the rest ``nn.Linear`` / ``nn.Dropout``."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..._lib import PATH_FUSED, PATH_ONLY_FWD, PATH_ONLY_INV
from ...autograd import AggregateFn
from .base_gnn import BaseGNNLayer

VERY_SMALL_NUMBER = 1e-10
VERY_NEG_NUMBER = -100000000000


class NSMBaseLayer(BaseGNNLayer):
    """Reference: nsm_gnn.py:13-78."""

    _direction = 0          # 0: head -> tail (NSMLayer), 1: tail -> head (NSMLayer_back)

    def __init__(self, args, num_entity, num_relation, entity_dim):
        super().__init__(args, num_entity, num_relation)
        self.num_entity = num_entity
        self.num_relation = num_relation
        self.entity_dim = entity_dim
        self.num_steps = args["num_step"]
        self.reason_kb = args["reason_kb"]
        self.init_layers(args)
        self._ws = ops.LayerWorkspace()

    def init_layers(self, args):
        D = self.entity_dim
        self.softmax_d1 = nn.Softmax(dim=1)
        self.score_func = nn.Linear(in_features=D, out_features=1)
        self.lin = nn.Linear(in_features=2 * D, out_features=D)            # unused, state_dict parity
        self.linear_dropout = args["linear_dropout"]
        self.linear_drop = nn.Dropout(p=self.linear_dropout)
        for i in range(self.num_steps):
            self.add_module("rel_linear" + str(i), nn.Linear(in_features=D, out_features=D))
            self.add_module("e2e_linear" + str(i), nn.Linear(in_features=2 * D, out_features=D))

    def init_reason(self, local_entity, kb_adj_mat, local_entity_emb, rel_features, query_node_emb=None):
        batch_size, max_local_entity = local_entity.size()
        self.local_entity_mask = (local_entity != self.num_entity).float()
        self.batch_size = batch_size
        self.max_local_entity = max_local_entity
        self.edge_list = kb_adj_mat
        self.rel_features = rel_features
        self.local_entity_emb = local_entity_emb
        self.num_relation = self.rel_features.size(0)
        self.possible_cand = []
        self.build_matrix()

    def _relation_features(self):
        # NSMLayer_back reads self.rel_features_inv (nsm_gnn.py:122), which the reference's init_reason
        # never sets: same AttributeError here unless the caller assigns it
        return self.rel_features if self._direction == 0 else self.rel_features_inv

    def _padded_e2e(self, e2e_linear):
        """[W_self | W_nbr | 0] (forward) / [W_self | 0 | W_nbr] (backward): the e2e_linear of a ReaRev layer
        with one instruction whose other direction contributes nothing."""
        D = self.entity_dim
        W = e2e_linear.weight
        zero = W.new_zeros(D, D)
        blocks = [W[:, :D], W[:, D:], zero] if self._direction == 0 else [W[:, :D], zero, W[:, D:]]
        return torch.cat(blocks, dim=1).contiguous()

    def _reach(self, current_dist):
        """sum over incoming facts of w_f * dist[src] per node (nsm_gnn.py:101-105 / :128-132): the fused
        walk over a [rel_total, 4] table of ones for this direction, zeros for the other."""
        plan = self.plan
        ones = torch.zeros((2, plan.rel_total, 4), dtype=torch.float32, device=current_dist.device)
        ones[self._direction] = 1.0
        return ops.aggregate_fused(plan, current_dist.detach().float(), ones)[:, 0].view(self.batch_size, -1)

    def forward(self, current_dist, relational_ins, step=0, return_score=False):
        """Reference: nsm_gnn.py:53-78."""
        B, N, D = self.batch_size, self.max_local_entity, self.entity_dim
        rel_linear = getattr(self, "rel_linear" + str(step))
        e2e_linear = getattr(self, "e2e_linear" + str(step))
        ins = relational_ins.squeeze(1).reshape(B, 1, D)
        relfeat = self._relation_features()
        answer_mask = self.local_entity_mask
        if self.reason_kb:
            answer_mask = answer_mask * (self._reach(current_dist) > VERY_SMALL_NUMBER).float()
        if torch.is_grad_enabled() or (self.training and self.linear_dropout > 0):
            T = rel_linear(relfeat.float())
            agg = AggregateFn.apply(self.plan, current_dist.float(), ins.float(), T, T)     # [BN, fwd | inv]
            nbr = agg[:, self._direction * D:(self._direction + 1) * D].reshape(B, N, D)
            nxt = torch.cat((self.local_entity_emb.float(), nbr), dim=2)
            self.local_entity_emb = F.relu(e2e_linear(self.linear_drop(nxt)))
            score_tp = self.score_func(self.linear_drop(self.local_entity_emb)).squeeze(dim=2)
            score_tp = score_tp + (1 - answer_mask) * VERY_NEG_NUMBER
            new_dist = self.softmax_d1(score_tp)
        else:
            h_out, score_tp, new_dist = ops.reason_layer(
                self.plan, self.local_entity_emb.detach().float(), current_dist.detach().float(),
                ins.detach().float(), relfeat.detach(), relfeat.detach(), rel_linear.weight, rel_linear.bias,
                self._padded_e2e(e2e_linear), e2e_linear.bias, self.score_func.weight, self.score_func.bias,
                answer_mask, ws=self._ws, path=PATH_FUSED | (PATH_ONLY_INV if self._direction else PATH_ONLY_FWD))
            self.local_entity_emb = h_out
        self.possible_cand.append(answer_mask)
        if return_score:
            return score_tp, new_dist
        return new_dist


class NSMLayer(NSMBaseLayer):
    """Forward reasoning along head -> tail (reference: nsm_gnn.py:83-112)."""
    _direction = 0


class NSMLayer_back(NSMBaseLayer):
    """Backward reasoning along tail -> head (reference: nsm_gnn.py:114-142)."""
    _direction = 1
