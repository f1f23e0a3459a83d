"""Drop-in for the reference's ``modules/kg_reasoning/reasongnn.py``.

Same class name, constructor, registered parameters (so released checkpoints load:
``rel_linear{i}``, ``e2e_linear{i}``, ``score_func``, ``pos_emb{i}``/``pos_emb_inv{i}``,
and the unused-but-present ``glob_lin``, ``lin``, ``lin_m`` - reference
``reasongnn.py:26-44``), same ``init_reason`` / ``forward`` signatures, return values and
side effects (``local_entity_emb``, ``local_entity_mask``, ``possible_cand``).  The body of
``forward`` is one call into libgnnrag_hip.so (``gnnrag_reason_layer``) under ``torch.no_grad()``.

With autograd enabled (``Trainer_KBQA.train_epoch``, train_model.py:222) the typed-edge
aggregation runs as an autograd function (HIP forward ``gnnrag_aggregate``, HIP backward
``gnnrag_aggregate_backward``); the dense projections and dropout around it are the same
``nn.Linear`` / ``nn.Dropout`` calls as the reference's, which autograd already knows.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...autograd import AggregateFn
from .base_gnn import BaseGNNLayer

VERY_NEG_NUMBER = -100000000000


class ReasonGNNLayer(BaseGNNLayer):
    """GNN reasoning layer of ReaRev on MI355X (reference: reasongnn.py:10-174)."""

    def __init__(self, args, num_entity, num_relation, entity_dim, alg):
        super().__init__(args, num_entity, num_relation)
        self.num_entity = num_entity
        self.num_relation = num_relation
        self.entity_dim = entity_dim
        self.alg = alg
        self.num_ins = args["num_ins"]
        self.num_gnn = args["num_gnn"]
        self.use_posemb = args["pos_emb"]
        self.init_layers(args)
        self._ws = ops.LayerWorkspace()
        # kernel path of the library (0 auto / 1 unfused / 2 fused); GNNRAG_PATH overrides for A/B runs
        self.path = int(os.environ.get("GNNRAG_PATH", "0"))
        # One library call per ReaRev iteration instead of one per layer (ops.LayerStack): the step-0 call runs all
        # num_gnn layers, the calls for steps 1.. hand out what it computed - provided the caller passes back the
        # distribution it was given and the same instructions, as ReaRev.forward does (rearev.py:208-210).
        self.use_stack = os.environ.get("GNNRAG_STACK", "1") != "0"
        self._stack = None
        self._ahead = None

    def init_layers(self, args):
        D = self.entity_dim
        if self.alg != "bfs":
            raise ValueError("ReasonGNNLayer supports alg='bfs' only (as the reference, reasongnn.py:33)")
        self.softmax_d1 = nn.Softmax(dim=1)
        self.score_func = nn.Linear(in_features=D, out_features=1)
        self.glob_lin = nn.Linear(in_features=D, out_features=D)          # unused, kept for state_dict parity
        self.lin = nn.Linear(in_features=2 * D, out_features=D)           # unused, kept for state_dict parity
        self.linear_dropout = args["linear_dropout"]
        self.linear_drop = nn.Dropout(p=self.linear_dropout)
        for i in range(self.num_gnn):
            self.add_module("rel_linear" + str(i), nn.Linear(in_features=D, out_features=D))
            self.add_module("e2e_linear" + str(i),
                            nn.Linear(in_features=2 * self.num_ins * D + D, out_features=D))
            if self.use_posemb:
                self.add_module("pos_emb" + str(i), nn.Embedding(self.num_relation, D))
                self.add_module("pos_emb_inv" + str(i), nn.Embedding(self.num_relation, D))
        self.lin_m = nn.Linear(in_features=self.num_ins * D, out_features=D)  # unused, state_dict parity

    def init_reason(self, local_entity, kb_adj_mat, local_entity_emb, rel_features, rel_features_inv,
                    query_entities, query_node_emb=None):
        batch_size, max_local_entity = local_entity.size()
        self.local_entity_mask = (local_entity != self.num_entity).float()
        self.batch_size = batch_size
        self.max_local_entity = max_local_entity
        self.edge_list = kb_adj_mat
        self.rel_features = rel_features
        self.rel_features_inv = rel_features_inv
        self.local_entity_emb = local_entity_emb
        self.num_relation = self.rel_features.size(0)
        self.possible_cand = []
        self.build_matrix()
        self.query_entities = query_entities
        self._stack = None             # bound to this batch's structure and relation features on first use
        self._ahead = None

    def forward(self, current_dist, relational_ins, step=0, return_score=False):
        """Next distribution and node representations (reference: reasongnn.py:134-174)."""
        if torch.is_grad_enabled() or (self.training and self.linear_dropout > 0):
            return self._forward_autograd(current_dist, relational_ins, step, return_score)
        if self.use_stack and self.num_gnn > 1:
            got = self._forward_stack(current_dist, relational_ins, step)
            if got is not None:
                h_out, score_tp, new_dist = got
                self.local_entity_emb = h_out
                self.possible_cand.append(self.local_entity_mask)
                return (score_tp, new_dist) if return_score else (new_dist, h_out)
        rel_linear = getattr(self, "rel_linear" + str(step))
        e2e_linear = getattr(self, "e2e_linear" + str(step))
        pos = pos_inv = None
        if self.use_posemb:
            pos = getattr(self, "pos_emb" + str(step)).weight
            pos_inv = getattr(self, "pos_emb_inv" + str(step)).weight
        B, N, D = self.batch_size, self.max_local_entity, self.entity_dim
        h_out, score_tp, new_dist = ops.reason_layer(
            self.plan, self.local_entity_emb.detach().float(), current_dist.detach().float(),
            relational_ins.detach().float(), self.rel_features.detach(), self.rel_features_inv.detach(),
            rel_linear.weight, rel_linear.bias, e2e_linear.weight, e2e_linear.bias,
            self.score_func.weight, self.score_func.bias, self.local_entity_mask,
            pos=pos, pos_inv=pos_inv, ws=self._ws, path=self.path)
        self.local_entity_emb = h_out
        self.possible_cand.append(self.local_entity_mask)
        if return_score:
            return score_tp, new_dist
        return new_dist, self.local_entity_emb

    def _forward_stack(self, current_dist, relational_ins, step):
        """(h, score, dist) of layer `step` from a whole-iteration run, or None when the call does not continue the
        sequence the step-0 call ran ahead (then the single-layer path computes it)."""
        if step == 0:
            if self._stack is None:
                layers = []
                for j in range(self.num_gnn):
                    rl, e2e = getattr(self, "rel_linear" + str(j)), getattr(self, "e2e_linear" + str(j))
                    pos = getattr(self, "pos_emb" + str(j)).weight if self.use_posemb else None
                    pos_inv = getattr(self, "pos_emb_inv" + str(j)).weight if self.use_posemb else None
                    layers.append((rl.weight, rl.bias, e2e.weight, e2e.bias, pos, pos_inv))
                self._stack = ops.LayerStack(self.plan, self.rel_features.detach().float(),
                                             self.rel_features_inv.detach().float(), layers, self.score_func.weight,
                                             self.score_func.bias, self.local_entity_mask, self.num_ins, path=self.path)
            h, score, dist = self._stack.run(self.local_entity_emb.detach().float(), current_dist.detach().float(),
                                             relational_ins.detach().float())
            self._ahead = dict(ins=relational_ins, h=h, score=score, dist=dist, given=[dist[j] for j in range(self.num_gnn)],
                               emb=[h[j] for j in range(self.num_gnn)])
            return self._ahead["emb"][0], score[0], self._ahead["given"][0]
        a = self._ahead
        if (a is None or step >= self.num_gnn or relational_ins is not a["ins"] or current_dist is not a["given"][step - 1]
                or self.local_entity_emb is not a["emb"][step - 1]):
            self._ahead = None
            return None
        return a["emb"][step], a["score"][step], a["given"][step]

    def _forward_autograd(self, current_dist, relational_ins, step, return_score):
        """Differentiable form of the same layer (reasongnn.py:134-174 op for op; the per-fact
        rel_linear / pos_emb of :71-79,98-105 applied per relation row, the four sparse products of
        :80-84,106-111 as ONE aggregation call with a HIP backward)."""
        rel_linear = getattr(self, "rel_linear" + str(step))
        e2e_linear = getattr(self, "e2e_linear" + str(step))
        B, N, D = self.batch_size, self.max_local_entity, self.entity_dim
        T_fwd = rel_linear(self.rel_features.float())
        T_inv = rel_linear(self.rel_features_inv.float())
        if self.use_posemb:
            pos = getattr(self, "pos_emb" + str(step)).weight
            pos_inv = getattr(self, "pos_emb_inv" + str(step)).weight
            pad = T_fwd.new_zeros(T_fwd.size(0) - pos.size(0), D)      # relation rows without a pos_emb row
            T_fwd = T_fwd + torch.cat([pos, pad], dim=0)
            T_inv = T_inv + torch.cat([pos_inv, pad], dim=0)
        agg = AggregateFn.apply(self.plan, current_dist.float(), relational_ins.float(), T_fwd, T_inv)
        state = torch.cat((self.local_entity_emb.float(), agg.view(B, N, -1)), dim=2)      # [h | fwd_0 | inv_0 | ...]
        self.local_entity_emb = F.relu(e2e_linear(self.linear_drop(state)))
        mask = self.local_entity_mask
        self.possible_cand.append(mask)
        score = self.score_func(self.linear_drop(self.local_entity_emb)).squeeze(dim=2) + (1 - mask) * VERY_NEG_NUMBER
        new_dist = self.softmax_d1(score)
        return (score, new_dist) if return_score else (new_dist, self.local_entity_emb)
