"""autograd bindings of the sparse operators.

Training (``Trainer_KBQA.train_epoch``, reference ``train_model.py:209-233``) differentiates
through ``reason_layer`` / ``reason_layer_inv`` (``reasongnn.py:61-116``) and ``TypeLayer``
(``layer_init.py:25-62``).  Here the typed-edge aggregation and its backward are HIP kernels
(``gnnrag_aggregate`` / ``gnnrag_aggregate_backward``, ``gnnrag_typelayer`` /
``gnnrag_typelayer_backward``); the dense projections around them stay ``nn.Linear`` calls, which
autograd already knows."""
from __future__ import annotations

import torch

from . import ops


class AggregateFn(torch.autograd.Function):
    """agg [BN, 2I*D] = typed-edge aggregation of all instructions, both directions."""

    @staticmethod
    def forward(ctx, plan, dist, ins, T_fwd, T_inv):
        dist = dist.detach().float().contiguous()
        ins = ins.detach().float().contiguous()
        T_fwd = T_fwd.detach().float().contiguous()
        T_inv = T_inv.detach().float().contiguous()
        ctx.plan = plan
        ctx.save_for_backward(dist, ins, T_fwd, T_inv)
        return ops.aggregate(plan, dist, ins, T_fwd, T_inv)

    @staticmethod
    def backward(ctx, g_agg):
        dist, ins, T_fwd, T_inv = ctx.saved_tensors
        g_dist, g_ins, g_Tf, g_Ti = ops.aggregate_backward(ctx.plan, dist, ins, T_fwd, T_inv,
                                                           g_agg.float().contiguous())
        return None, g_dist.view(ctx.plan.B, ctx.plan.N), g_ins, g_Tf, g_Ti


class TypeAggFn(torch.autograd.Function):
    """h0 [BN, D] = relu(sum over incident facts of v_f T[rel_f]) (both directions)."""

    @staticmethod
    def forward(ctx, plan, T, use_w_rel):
        T = T.detach().float().contiguous()
        h0 = ops.typelayer(plan, T, use_w_rel)
        ctx.plan = plan
        ctx.use_w_rel = use_w_rel
        ctx.save_for_backward(h0)
        return h0

    @staticmethod
    def backward(ctx, g_h0):
        (h0,) = ctx.saved_tensors
        g_pre = (g_h0.float() * (h0 > 0)).contiguous()
        return None, ops.typelayer_backward(ctx.plan, g_pre, ctx.use_w_rel), None
