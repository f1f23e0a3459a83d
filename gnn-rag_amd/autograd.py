"""autograd bindings of the sparse operators.

Training (``Trainer_KBQA.train_epoch``, reference ``train_model.py:209-233``) differentiates
through ``reason_layer`` / ``reason_layer_inv`` (``reasongnn.py:61-116``) and ``TypeLayer``
(``layer_init.py:25-62``).  Here the typed-edge aggregation and its backward are HIP kernels
(``gnnrag_aggregate`` / ``gnnrag_aggregate_backward``, ``gnnrag_typelayer`` /
``gnnrag_typelayer_backward``); the dense projections around them run on the library's matrix-core kernels in
both directions (:class:`LinearFn`: ``gnnrag_linear`` for y and dx, ``gnnrag_gemm_tn`` for dW)."""
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


class FusedAggregateFn(torch.autograd.Function):
    """nbr [BN, D] = the fused walk over per-question relation tables P [2, rel_total, D] (``gnnrag_aggregate_fused``);
    backward on ``gnnrag_aggregate_fused_backward`` (gather kernels, fixed summation order)."""

    @staticmethod
    def forward(ctx, plan, dist, P):
        dist = dist.detach().float().contiguous()
        P = P.detach().float().contiguous()
        ctx.plan = plan
        ctx.save_for_backward(dist, P)
        return ops.aggregate_fused(plan, dist, P)

    @staticmethod
    def backward(ctx, g_nbr):
        dist, P = ctx.saved_tensors
        g_dist, g_P = ops.aggregate_fused_backward(ctx.plan, dist, P, g_nbr.float().contiguous())
        return None, g_dist.view(ctx.plan.B, ctx.plan.N), g_P


def relation_tables_dense(plan, T_fwd, T_inv, ins, W_e2e):
    """Differentiable per-question relation tables of the fused form in plain torch ops:
    P[d, (b, r), :] = sum_i W_e2e[:, (1 + 2 i + d) D : (2 + 2 i + d) D] . relu(T_d[r] * ins[b, i]) over the compact rows
    (question, relation in use) of the structure - a few ten thousand rows, where the per-fact form of the reference
    (reasongnn.py:71-79) has one row per fact.  [2, rel_total, D]."""
    rows = plan.rel_rows_device()
    b, r = rows[:, 0].long(), rows[:, 1].long()
    I, D = ins.shape[1], ins.shape[2]
    q = ins.index_select(0, b)                                             # [M, I, D]
    out = []
    for d, T in enumerate((T_fwd, T_inv)):
        z = torch.relu(T.index_select(0, r).unsqueeze(1) * q)             # [M, I, D]
        Wd = torch.stack([W_e2e[:, (1 + 2 * i + d) * D:(2 + 2 * i + d) * D] for i in range(I)])    # [I, D_out, D]
        out.append(torch.einsum("mik,iok->mo", z, Wd))
    return torch.stack(out)


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


class LinearFn(torch.autograd.Function):
    """``y = act(x W^T + b)`` (``nn.Linear`` + optional ReLU) on the hand-written kernels, forward and backward:
    forward and ``dx = dy W`` are ``gnnrag_linear`` calls (the second with the transposed weight, a [K, Nout] copy of
    a few hundred KB), ``dW = dy^T x`` is ``gnnrag_gemm_tn``; ``db`` is a column sum.  x: [M, K] fp32 contiguous."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        x = x.detach().float().contiguous()
        Wd = W.detach().float().contiguous()
        y = ops.linear(x, Wd, None if b is None else b.detach().float().contiguous(), relu=relu)
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, Wd, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W, y = ctx.saved_tensors
        g = gy.float()
        if ctx.relu:
            g = g * (y > 0)
        g = g.contiguous()
        gx = ops.linear(g, W.t().contiguous()) if ctx.needs_input_grad[0] else None
        gW = ops.gemm_tn(g, x) if ctx.needs_input_grad[1] else None
        gb = g.sum(dim=0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gW, gb, None


def linear(x: torch.Tensor, W: torch.Tensor, b=None, relu: bool = False) -> torch.Tensor:
    """Differentiable ``act(x W^T + b)`` over the last dimension of x on :class:`LinearFn`."""
    shp = x.shape
    y = LinearFn.apply(x.reshape(-1, shp[-1]), W, b, relu)
    return y.view(*shp[:-1], W.shape[0])
