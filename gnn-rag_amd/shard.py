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

import os

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


def balanced_ranges(weights, world: int):
    """Contiguous shards [lo, hi) of len(weights) questions whose summed weights are as even as a contiguous split
    allows ("for load balance on real data, shard by cumulative F_g not by count" - WebQSP
    subgraphs differ by two orders of magnitude in size, and the slowest rank sets the step).  Greedy sweep against
    the ideal cumulative targets; every rank gets at least one question while questions remain."""
    w = np.maximum(np.asarray(weights, dtype=np.float64), 0.0) + 1e-9          # empty questions still cost a slot
    B = len(w)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    out, lo = [], 0
    for r in range(world):
        left = world - r - 1                                                   # ranks after this one
        if r == world - 1:
            hi = B
        else:
            target = cum[-1] * (r + 1) / world
            hi = int(np.searchsorted(cum, target, side="left"))
            # the boundary question goes to the side that leaves the smaller error
            if hi > lo + 1 and hi <= B and abs(cum[hi - 1] - target) <= abs(cum[hi] - target if hi < len(cum) else np.inf):
                hi -= 1
            hi = max(hi, min(lo + 1, B))                                       # at least one question if any are left
            hi = min(hi, max(B - left, lo))                                    # leave one for each later rank
        out.append((lo, hi))
        lo = hi
    return out


def facts_per_question(edge_tuple, B: int) -> np.ndarray:
    """F_g: facts (typed edges + self loops) of every question of a batch tuple."""
    known = getattr(edge_tuple, "facts_per_question", None)       # data/fact_mat.ShardedFacts: counted, not built
    if known is not None:
        return np.asarray(known)[:B]
    return np.bincount(np.asarray(edge_tuple[3]).astype(np.int64), minlength=B)[:B]


def shard_edge_tuple(edge_tuple, N: int, lo: int, hi: int):
    """Facts of questions [lo, hi), re-based so the shard is a self-contained batch.
    Facts of one question are contiguous and ``batch_ids`` is non-decreasing
    (``_build_fact_mat``, dataset_load.py:481-506)."""
    if hasattr(type(edge_tuple), "shard"):         # data/fact_mat.BatchFacts: the id arrays live on the GPU (the type is
                                                   # asked, not the lazy `hrt_device` property, which would build the block)
        return edge_tuple.shard(lo, hi)
    heads, rels, tails, bids, _, wl, wrl = edge_tuple
    bids = np.asarray(bids)
    if len(bids) and (np.diff(bids) < 0).any():
        raise ValueError("batch_ids must be non-decreasing (facts grouped by question)")
    a = int(np.searchsorted(bids, lo, side="left"))
    b = int(np.searchsorted(bids, hi, side="left"))
    off = lo * N
    return (np.asarray(heads)[a:b] - off, np.asarray(rels)[a:b], np.asarray(tails)[a:b] - off,
            bids[a:b] - lo, np.arange(b - a, dtype=np.int64), list(wl[a:b]), list(wrl[a:b]))


def shard_ranges(batch: tuple, world: int, balance: str = "facts"):
    """The [lo, hi) question range of every rank for one batch: ``balance="facts"`` evens out the facts per rank
    (every rank derives the same split from the batch tuple it already holds - no communication), ``"count"`` the
    number of questions."""
    B = batch[0].shape[0]
    decided = getattr(batch[2], "ranges", None)        # data/fact_mat.ShardedFacts: the loader already split this batch
    if decided is not None and len(decided) == world:
        return [tuple(r) for r in decided]
    if balance == "count":
        return [question_range(B, r, world) for r in range(world)]
    if balance != "facts":
        raise ValueError("balance must be 'facts' or 'count'")
    return balanced_ranges(facts_per_question(batch[2], B), world)


def shard_batch(batch: tuple, rank: int, world: int, balance: str = "count") -> tuple:
    """Shards a reference batch tuple (``get_batch`` output, dataset_load.py:613-629):
    ``(local_entity, query_entities, kb_adj_mat, query_text, seed_dist, true_batch_id,
    answer_dist[, answer_lists])``."""
    local_entity = batch[0]
    B, N = local_entity.shape
    lo, hi = shard_ranges(batch, world, balance)[rank]
    out = [batch[0][lo:hi], batch[1][lo:hi], shard_edge_tuple(batch[2], N, lo, hi), batch[3][lo:hi],
           batch[4][lo:hi], batch[5], batch[6][lo:hi]]
    if len(batch) > 7:
        out.append(batch[7][lo:hi])
    return tuple(out)


def gather_rows_async(local: torch.Tensor, B: int, group: Optional[dist.ProcessGroup] = None, ranges=None):
    """:func:`gather_rows` split in two: the collective is ENQUEUED (``async_op=True``: RCCL runs it on its own stream
    behind an event on the caller's) and a ``finish()`` callable is returned that makes the caller's stream wait for it
    and assembles ``[B, ...]``.  Lets a loop over batches overlap a batch's all-gather with the next batch's kernels:
    the distribution of batch k is only read after batch k + 1 has been enqueued."""
    if not dist.is_available() or not dist.is_initialized():
        if local.shape[0] != B:
            raise ValueError("not distributed, but the local shard is not the whole batch")
        return lambda: local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if ranges is None:
        ranges = [question_range(B, r, world) for r in range(world)]
    lo, hi = ranges[rank]
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], hi - lo))
    rows = max(h - l for l, h in ranges)
    tail = tuple(local.shape[1:])
    if hi - lo == rows:
        send = local.contiguous()                        # full-size shard: it goes out as it is
    else:
        send = local.new_zeros((rows,) + tail)
        send[: hi - lo] = local
    recv = local.new_empty((world * rows,) + tail)
    work = dist.all_gather_into_tensor(recv, send, group=group, async_op=True)

    def finish():
        work.wait()
        if all(h - l == rows for l, h in ranges):
            return recv
        return torch.cat([recv[r * rows: r * rows + (h - l)] for r, (l, h) in enumerate(ranges)], dim=0)

    finish._keep = (send, recv)                          # the buffers stay alive until the collective has run
    return finish


def gather_rows(local: torch.Tensor, B: int, group: Optional[dist.ProcessGroup] = None,
                ranges=None) -> torch.Tensor:
    """All-gathers per-question rows ``[b_local, ...]`` of contiguous shards into ``[B, ...]``.
    One collective (``all_gather_into_tensor``; RCCL on GPUs, gloo in the CPU tests);
    shards are padded to the largest shard so every rank contributes the same count.  ``ranges``: the [lo, hi) of
    every rank (``shard_ranges``) when the split is not the even one."""
    return gather_rows_async(local, B, group, ranges)()


def shard_model(model, group: Optional[dist.ProcessGroup] = None, balance: str = "facts"):
    """Question-sharded evaluation of a reference model (``ReaRev`` / ``NSM``): ``model(batch)`` then runs
    this rank's contiguous question range only and all-gathers the scored nodes, so that every rank hands
    the full ``pred_dist [B, N]`` to the unchanged ``Evaluator`` (BASELINE config "dev set batched across
    8 GPUs, question-sharded, RCCL gather").  Returns the same 4-tuple as ``Model.forward``
    (rearev.py:243): the loss is the batch mean (per-rank means weighted by their question counts,
    all-reduced), ``pred`` the argmax of the gathered distribution.  Training calls, single-process runs and
    batches with fewer questions than ranks pass through unchanged.  ``balance``: "facts" splits the batch so that
    every rank gets about the same number of facts (the slowest rank sets the step), "count" by question count."""
    inner = model.forward

    def forward(batch, training=False):
        if training or not dist.is_available() or not dist.is_initialized():
            return inner(batch, training=training)
        world = dist.get_world_size(group)
        B = batch[0].shape[0]
        # (GNNRAG_FORCE_DIST=1: a single rank still shards - trivially - and runs both collectives, so that a
        # 1-GPU box exercises the RCCL path)
        if (world == 1 and os.environ.get("GNNRAG_FORCE_DIST") != "1") or B < world:
            return inner(batch, training=training)
        rank = dist.get_rank(group)
        ranges = shard_ranges(batch, world, balance)
        lo, hi = ranges[rank]
        loss, _, pred_dist, tp_list = inner(shard_batch(batch, rank, world, balance), training=training)
        full = gather_rows(pred_dist, B, group, ranges=ranges)
        total = loss.detach().float().reshape(1) * float(hi - lo)       # calc_loss_label divides by the local batch size
        dist.all_reduce(total, group=group)
        return total[0] / B, torch.max(full, dim=1)[1], full, tp_list

    model.forward = forward
    return model


def sharded_training_step(model, batch: tuple, group: Optional[dist.ProcessGroup] = None, balance: str = "facts"):
    """Data-parallel training step over question shards (SURVEY.md section 8e "Training (later): all-reduce of
    grads"): this rank runs ``model(shard, training=True)`` on its contiguous question range, scales the loss by its
    share of the batch (``calc_loss_label`` divides by the LOCAL batch size, rearev.py:156-160), back-propagates, and
    the parameter gradients of all ranks are summed with ONE all-reduce of a flat buffer (~2.5 MB at D = 200: latency
    bound on xGMI, so one bucket) - afterwards every rank holds the gradient of the whole-batch loss and the caller's
    optimizer step keeps the replicas identical.  Dropout / fact dropout draw from each rank's own RNG streams, as in
    any data-parallel run.  Returns (batch-mean loss, local model outputs).  Single process: the plain step.

    Gradient accumulation: whatever a parameter's ``.grad`` held on entry is kept as it is and only THIS step's
    contribution is all-reduced and added to it (the step back-propagates into cleared ``.grad`` fields and restores
    the earlier content afterwards) - summing stale contributions over the ranks would inflate them world-size times."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        out = model(batch, training=True)
        out[0].backward()
        return out[0].detach(), out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = batch[0].shape[0]
    ranges = shard_ranges(batch, world, balance)
    lo, hi = ranges[rank]
    params = [p for p in model.parameters() if p.requires_grad]
    earlier = [p.grad for p in params]                      # accumulated by the caller before this step
    for p in params:
        p.grad = None
    if hi > lo:
        out = model(shard_batch(batch, rank, world, balance), training=True)
        local = out[0] * (float(hi - lo) / B)
        local.backward()
        loss_part = local.detach().float().reshape(1)
    else:                                                   # more ranks than questions: contributes zeros
        out, loss_part = None, torch.zeros(1, device=params[0].device)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params] + [loss_part])
    dist.all_reduce(flat, group=group)
    off = 0
    for p, old in zip(params, earlier):
        n = p.numel()
        g = flat[off: off + n].view_as(p)
        if old is None:
            p.grad = g.clone()
        else:
            p.grad = old.add_(g)
        off += n
    return flat[off], out
