"""Question-sharded data parallelism (one process per GPU, RCCL over xGMI).

Every question is its own disconnected subgraph (node ids are offset by ``i * N``,
reference ``gnn/dataset_load.py:483``), the softmax is per question
(``reasongnn.py:169``) and parameters are read-only, so a batch shards by whole
questions with NO exchange inside the forward.  The only collective is one all-gather of
the scored nodes (``pred_dist`` shards ``[B/G, N]`` fp32) so that every rank / rank 0
can hand the full ``[B, N]`` to the unchanged ``Evaluator``.  Messages are tiny
(<= 2.6 MB per rank at 32 x 20k), i.e. latency-bound on the xGMI mesh.

Because a destination node's facts are summed in ascending fact id regardless of which
rank owns the question, the gathered result is bit-identical to the single-GPU result.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def question_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of B questions; sizes differ by at most one."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_edge_tuple(edge_tuple, N: int, lo: int, hi: int):
    """Facts of questions [lo, hi), re-based so the shard is a self-contained batch.
    Facts of one question are contiguous and ``batch_ids`` is non-decreasing
    (``_build_fact_mat``, dataset_load.py:481-506)."""
    heads, rels, tails, bids, _, wl, wrl = edge_tuple
    bids = np.asarray(bids)
    if len(bids) and (np.diff(bids) < 0).any():
        raise ValueError("batch_ids must be non-decreasing (facts grouped by question)")
    a = int(np.searchsorted(bids, lo, side="left"))
    b = int(np.searchsorted(bids, hi, side="left"))
    off = lo * N
    return (np.asarray(heads)[a:b] - off, np.asarray(rels)[a:b], np.asarray(tails)[a:b] - off,
            bids[a:b] - lo, np.arange(b - a, dtype=np.int64), list(wl[a:b]), list(wrl[a:b]))


def shard_batch(batch: tuple, rank: int, world: int) -> tuple:
    """Shards a reference batch tuple (``get_batch`` output, dataset_load.py:613-629):
    ``(local_entity, query_entities, kb_adj_mat, query_text, seed_dist, true_batch_id,
    answer_dist[, answer_lists])``."""
    local_entity = batch[0]
    B, N = local_entity.shape
    lo, hi = question_range(B, rank, world)
    out = [batch[0][lo:hi], batch[1][lo:hi], shard_edge_tuple(batch[2], N, lo, hi), batch[3][lo:hi],
           batch[4][lo:hi], batch[5], batch[6][lo:hi]]
    if len(batch) > 7:
        out.append(batch[7][lo:hi])
    return tuple(out)


def gather_rows(local: torch.Tensor, B: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """All-gathers per-question rows ``[b_local, ...]`` of contiguous shards into ``[B, ...]``.
    One collective (``all_gather_into_tensor``; RCCL on GPUs, gloo in the CPU tests);
    shards are padded to the largest shard so every rank contributes the same count."""
    if not dist.is_available() or not dist.is_initialized():
        if local.shape[0] != B:
            raise ValueError("not distributed, but the local shard is not the whole batch")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = question_range(B, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], hi - lo))
    rows = (B + world - 1) // world
    tail = tuple(local.shape[1:])
    if hi - lo == rows:
        send = local.contiguous()                        # even split: the shard goes out as it is
    else:
        send = local.new_zeros((rows,) + tail)
        send[: hi - lo] = local
    recv = local.new_empty((world * rows,) + tail)
    dist.all_gather_into_tensor(recv, send, group=group)
    if B % world == 0:
        return recv
    parts = []
    for r in range(world):
        l, h = question_range(B, r, world)
        parts.append(recv[r * rows: r * rows + (h - l)])
    return torch.cat(parts, dim=0)


def shard_model(model, group: Optional[dist.ProcessGroup] = None):
    """Question-sharded evaluation of a reference model (``ReaRev`` / ``NSM``): ``model(batch)`` then runs
    this rank's contiguous question range only and all-gathers the scored nodes, so that every rank hands
    the full ``pred_dist [B, N]`` to the unchanged ``Evaluator`` (BASELINE config "dev set batched across
    8 GPUs, question-sharded, RCCL gather").  Returns the same 4-tuple as ``Model.forward``
    (rearev.py:243): the loss is the batch mean (per-rank means weighted by their question counts,
    all-reduced), ``pred`` the argmax of the gathered distribution.  Training calls, single-process runs and
    batches with fewer questions than ranks pass through unchanged."""
    inner = model.forward

    def forward(batch, training=False):
        if training or not dist.is_available() or not dist.is_initialized():
            return inner(batch, training=training)
        world = dist.get_world_size(group)
        B = batch[0].shape[0]
        if world == 1 or B < world:
            return inner(batch, training=training)
        rank = dist.get_rank(group)
        lo, hi = question_range(B, rank, world)
        loss, _, pred_dist, tp_list = inner(shard_batch(batch, rank, world), training=training)
        full = gather_rows(pred_dist, B, group)
        total = loss.detach().float().reshape(1) * float(hi - lo)       # calc_loss_label divides by the local batch size
        dist.all_reduce(total, group=group)
        return total[0] / B, torch.max(full, dim=1)[1], full, tp_list

    model.forward = forward
    return model
