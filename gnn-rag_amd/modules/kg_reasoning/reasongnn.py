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
from ...autograd import AggregateFn, FusedAggregateFn, linear as ag_linear, relation_tables_dense
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
        # The step-0 call of every ReaRev iteration starts from the seed distribution (rearev.py:208), whose support is
        # the question's seeds: the library then computes relation tables and neighbour sums for the seeds' frontier
        # only (GNNRAG_PATH_SEED_PRIOR; derived from the distribution on the device, so ANY prior gives the same
        # result - a dense one just slowly).  GNNRAG_SEED_PRIOR=0 switches the hint off.
        self.seed_prior = os.environ.get("GNNRAG_SEED_PRIOR", "1") != "0"
        # One library call per ReaRev iteration instead of one per layer (ops.LayerStack): the step-0 call runs all
        # num_gnn layers, the calls for steps 1.. hand out what it computed - provided the caller passes back the
        # distribution it was given and the same instructions, as ReaRev.forward does (rearev.py:208-210).
        self.use_stack = os.environ.get("GNNRAG_STACK", "1") != "0"
        self._stack = None
        self._ahead = None
        # hipGraph replay of the whole-iteration call for SMALL batches (B * N < 4096 node slots, e.g. one WebQSP question:
        # ~50 launches of 3-14 us per iteration are bound by the host's launch rate): iteration 1 of a forward runs eager
        # (it also computes the relation projections), iteration 2 captures the sequence over fixed buffers, iterations
        # 2..T replay it.  GNNRAG_GRAPH=1 switches it on, 0 off (default: off - a capture per batch costs about what it
        # saves at num_iter = 3; bit-identical to eager either way).
        self.graph_small = os.environ.get("GNNRAG_GRAPH", "0") == "1"
        # Hidden sizes that are not a multiple of 4 (the released checkpoints use entity_dim 50) would send every
        # kernel down its scalar path and the walk to the table-row gather.  The inference path therefore works on
        # ZERO-PADDED copies (D -> next multiple of 8): padded parameter rows / columns are zero, so the extra
        # columns of every intermediate stay exactly 0 and the first D columns are what the unpadded computation
        # gives; the node state is kept padded between calls and handed to the caller as a [:, :, :D] view.
        self.pad_dim = os.environ.get("GNNRAG_PAD_DIM", "1") != "0"
        # training: dense projections on the library's kernels forward and backward (0: nn.Linear / rocBLAS)
        self.native_dense = os.environ.get("GNNRAG_NATIVE_DENSE", "1") != "0"
        # training without active dropout: the fused form (relation tables + fused walk with its own backward)
        self.train_fused = os.environ.get("GNNRAG_TRAIN_FUSED", "1") != "0"
        self._padded = None

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
        self._stack_key = None
        self._padded = None            # padded copies hold this batch's relation features
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
        P = self._inference_params()
        W_rel, b_rel, W_e2e, b_e2e, pos, pos_inv = P["layers"][step]
        h_out, score_tp, new_dist = ops.reason_layer(
            self.plan, self._state_in(P), current_dist.detach().float(), self._pad_last(relational_ins.detach().float(), P),
            P["relfeat"], P["relfeat_inv"], W_rel, b_rel, W_e2e, b_e2e, P["w_score"], P["b_score"],
            self.local_entity_mask, pos=pos, pos_inv=pos_inv, ws=self._ws, path=self._path_of(step))
        self.local_entity_emb = self._state_out(h_out, P)
        self.possible_cand.append(self.local_entity_mask)
        if return_score:
            return score_tp, new_dist
        return new_dist, self.local_entity_emb

    def _path_of(self, step):
        from ..._lib import PATH_SEED_PRIOR
        return self.path | (PATH_SEED_PRIOR if (self.seed_prior and step == 0) else 0)

    # ---- zero-padded inference copies (hidden size not a multiple of 4) ------------------------------------
    def _inference_params(self):
        """Parameters and relation features as the kernels get them: the module's own tensors, or - when the hidden
        size is padded - zero-padded copies, rebuilt when a parameter changes (checked by storage and version)."""
        D, I = self.entity_dim, self.num_ins
        Dp = D if (D % 4 == 0 or not self.pad_dim) else (D + 7) // 8 * 8
        mods = [(getattr(self, "rel_linear" + str(j)), getattr(self, "e2e_linear" + str(j))) for j in range(self.num_gnn)]
        src = [self.score_func.weight, self.score_func.bias]
        for rl, e2e in mods:
            src += [rl.weight, rl.bias, e2e.weight, e2e.bias]
        if self.use_posemb:
            for j in range(self.num_gnn):
                src += [getattr(self, "pos_emb" + str(j)).weight, getattr(self, "pos_emb_inv" + str(j)).weight]
        wkey = (Dp,) + tuple((t.data_ptr(), t._version) for t in src)
        key = (id(self.rel_features), id(self.rel_features_inv)) + wkey
        if self._padded is not None and self._padded["key"] == key:
            return self._padded

        def pad_cols(t):                     # [.., D] -> [.., Dp]
            return t if Dp == D else F.pad(t.detach().float(), (0, Dp - D))

        def pad_sq(t):                       # [D, D] -> [Dp, Dp]
            return t if Dp == D else F.pad(t.detach().float(), (0, Dp - D, 0, Dp - D))

        # the parameters' copies outlive a batch (an evaluation run never changes them: 18 pads per forward at D = 50
        # otherwise); only the relation features' copies follow the batch
        W = getattr(self, "_padded_w", None)
        if W is None or W["wkey"] != wkey:
            layers = []
            for j, (rl, e2e) in enumerate(mods):
                W_e2e = e2e.weight
                if Dp != D:                  # every [D, D] column block of e2e_linear.weight moves to its padded place
                    W_e2e = torch.cat([pad_sq(e2e.weight[:, k * D:(k + 1) * D]) for k in range(2 * I + 1)], dim=1).contiguous()
                pos = pos_inv = None
                if self.use_posemb:
                    pos = pad_cols(getattr(self, "pos_emb" + str(j)).weight).contiguous()
                    pos_inv = pad_cols(getattr(self, "pos_emb_inv" + str(j)).weight).contiguous()
                layers.append((pad_sq(rl.weight).contiguous(), pad_cols(rl.bias).contiguous(), W_e2e,
                               pad_cols(e2e.bias).contiguous(), pos, pos_inv))
            W = self._padded_w = dict(wkey=wkey, layers=layers,
                                      w_score=pad_cols(self.score_func.weight.reshape(-1)).contiguous(),
                                      b_score=self.score_func.bias)
        self._padded = dict(key=key, D=D, Dp=Dp, layers=W["layers"],
                            relfeat=pad_cols(self.rel_features.detach().float()).contiguous(),
                            relfeat_inv=pad_cols(self.rel_features_inv.detach().float()).contiguous(),
                            w_score=W["w_score"], b_score=W["b_score"])
        return self._padded

    @staticmethod
    def _pad_last(t, P):
        return t if P["Dp"] == P["D"] else F.pad(t, (0, P["Dp"] - P["D"]))

    def _state_in(self, P):
        """Node state as the kernels read it: the padded tensor behind the view handed out by the previous call, or
        a padded copy of whatever the caller installed (TypeLayer's output, rearev.py:140-153)."""
        h = self.local_entity_emb
        if P["Dp"] == P["D"]:
            return h.detach().float()
        base = getattr(h, "_gnnrag_padded", None)
        if base is not None and base.shape[-1] == P["Dp"]:
            return base
        return self._pad_last(h.detach().float(), P)

    @staticmethod
    def _state_out(h_pad, P):
        """[.., Dp] kernel output -> what the caller sees: a [.., :D] view that remembers its padded base."""
        if P["Dp"] == P["D"]:
            return h_pad
        view = h_pad[..., : P["D"]]
        view._gnnrag_padded = h_pad
        return view

    def _forward_stack(self, current_dist, relational_ins, step):
        """(h, score, dist) of layer `step` from a whole-iteration run, or None when the call does not continue the
        sequence the step-0 call ran ahead (then the single-layer path computes it)."""
        if step == 0:
            P = self._inference_params()
            if self._stack is None or self._stack_key != P["key"]:
                self._stack = ops.LayerStack(self.plan, P["relfeat"], P["relfeat_inv"], P["layers"], P["w_score"],
                                             P["b_score"], self.local_entity_mask, self.num_ins, path=self._path_of(0))
                self._stack_key = P["key"]
            st = self._stack
            dist_in = current_dist.detach().float()
            ins_in = self._pad_last(relational_ins.detach().float(), P)
            if (self.graph_small and st._proj_valid and self.batch_size * self.max_local_entity < 4096 and st.L >= 2
                    and dist_in.is_contiguous()):
                if getattr(st, "_graph_rest", None) is None or st._graph_in[0].data_ptr() != dist_in.data_ptr():
                    st.capture_rest(self._state_in(P), dist_in, ins_in)           # iteration 2: capture ...
                h, score, dist = st.replay_rest(self._state_in(P), dist_in, ins_in)   # ... and replay (iterations 2..T)
                # the captured sequence writes fixed buffers: the distributions handed to the caller are copies (the
                # reference keeps every iteration's distribution, rearev.py:211); the node state is read in place by
                # the next replay
                score, dist = score.clone(), dist.clone()
            else:
                h, score, dist = st.run(self._state_in(P), dist_in, ins_in)
            self._ahead = dict(ins=relational_ins, h=h, score=score, dist=dist, given=[dist[j] for j in range(self.num_gnn)],
                               emb=[self._state_out(h[j], P) for j in range(self.num_gnn)])
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
        # the dense projections on the library's kernels in both directions (autograd.LinearFn) when their float4 /
        # 16-byte requirements hold; nn.Linear (rocBLAS) otherwise
        native = self.native_dense and D % 4 == 0 and self.rel_features.is_cuda
        lin = (lambda x, m, relu=False: ag_linear(x, m.weight, m.bias, relu)) if native else \
              (lambda x, m, relu=False: F.relu(m(x)) if relu else m(x))
        T_fwd = lin(self.rel_features.float(), rel_linear)
        T_inv = lin(self.rel_features_inv.float(), rel_linear)
        if self.use_posemb:
            pos = getattr(self, "pos_emb" + str(step)).weight
            pos_inv = getattr(self, "pos_emb_inv" + str(step)).weight
            pad = T_fwd.new_zeros(T_fwd.size(0) - pos.size(0), D)      # relation rows without a pos_emb row
            T_fwd = T_fwd + torch.cat([pos, pad], dim=0)
            T_inv = T_inv + torch.cat([pos_inv, pad], dim=0)
        # Training on the FUSED form (round 4): without dropout between the aggregation and e2e_linear
        # (reasongnn.py:161-163; dropout acts on cat(h, agg) elementwise and cannot be pushed through the
        # re-association) the neighbour part is e2e_linear's column blocks applied to per-question relation tables
        # (a dense, differentiable expression over a few ten thousand rows) and the fused walk with its own backward:
        # agg [BN, 2I D] and the K = 2I D product never exist.  GNNRAG_TRAIN_FUSED=0 keeps the unfused autograd form.
        drop_active = self.training and self.linear_dropout > 0
        # (the fused walk's backward gathers whole rows: D <= 1024; beyond that the unfused form below keeps its LDS-sum backward)
        if native and self.train_fused and not drop_active and self.plan.rel_total > 0 and D <= 1024:
            W = e2e_linear.weight
            P = relation_tables_dense(self.plan, T_fwd, T_inv, relational_ins.float(), W)
            nbr = FusedAggregateFn.apply(self.plan, current_dist.float(), P)
            pre = ag_linear(self.local_entity_emb.float().reshape(B * N, D), W[:, :D], e2e_linear.bias) + nbr
            self.local_entity_emb = F.relu(pre).view(B, N, D)
            mask = self.local_entity_mask
            self.possible_cand.append(mask)
            score = self.score_func(self.local_entity_emb).squeeze(dim=2) + (1 - mask) * VERY_NEG_NUMBER
            new_dist = self.softmax_d1(score)
            return (score, new_dist) if return_score else (new_dist, self.local_entity_emb)
        agg = AggregateFn.apply(self.plan, current_dist.float(), relational_ins.float(), T_fwd, T_inv)
        if native:
            # e2e_linear over cat(h, agg) as two products (no [BN, (2I+1)D] copy of the concatenation): dropout acts
            # elementwise, so dropping the two parts separately is the same operator (reasongnn.py:161-163)
            W = e2e_linear.weight
            pre = ag_linear(self.linear_drop(self.local_entity_emb.float()).reshape(B * N, D), W[:, :D], e2e_linear.bias) \
                + ag_linear(self.linear_drop(agg), W[:, D:], None)
            self.local_entity_emb = F.relu(pre).view(B, N, D)
        else:
            state = torch.cat((self.local_entity_emb.float(), agg.view(B, N, -1)), dim=2)  # [h | fwd_0 | inv_0 | ...]
            self.local_entity_emb = F.relu(e2e_linear(self.linear_drop(state)))
        mask = self.local_entity_mask
        self.possible_cand.append(mask)
        score = self.score_func(self.linear_drop(self.local_entity_emb)).squeeze(dim=2) + (1 - mask) * VERY_NEG_NUMBER
        new_dist = self.softmax_d1(score)
        return (score, new_dist) if return_score else (new_dist, self.local_entity_emb)
