"""Evaluator tail on the device.

``Evaluator.evaluate`` (reference ``gnn/evaluate.py:147-240``) converts, per question, N probabilities,
N entity ids and N seed flags to Python lists and filters them in a Python loop (``:188-207``); at
64 x 2000 slots per batch that costs more than the GPU forward.  :func:`patch_evaluator` makes the reference's OWN
``evaluate`` method run on batches whose candidate axis has been compacted on the device: one kernel
(``gnnrag_topp_candidates``: the filter of ``:198-205``, the stable sort by probability and the top-p cut of
``f1_and_hits``, ``:34-51``) and one small D2H hand it, per question, exactly the slots it would retrieve, best first,
padded with the pad entity.  Everything else - metrics, ``.info`` records, prints, control flow - IS the reference's
code (nothing of it is restated here): its loop now walks a few retrieved slots instead of N.

    from gnnrag_amd.eval_tail import patch_evaluator
    patch_evaluator(trainer.evaluator)          # the reference file is untouched
"""
from __future__ import annotations

import os
import types

import numpy as np
import torch

from . import ops


# what the compacted batches of this process held (tools/run_reference.py prints it with its stage timers): the Evaluator's
# tail is the reference's own per-candidate Python, so its cost follows `retrieved`, not the subgraph width
STATS = {"questions": 0, "retrieved": 0, "retrieved_max": 0}

TOPP_MAX_N = 1 << 24    # the kernel takes any N (questions with > 16384 slots filter first and sort the survivors in a
                        # workspace); beyond 2^24 slots the batch is left as it is and the reference's own loop walks it


def _retrieved_arrays(pred_dist: torch.Tensor, local_entity: np.ndarray, query_entities: np.ndarray, pad_ent_id: int,
                      ignore_prob: float, eps: float):
    """One kernel, two small device -> host copies: (entity ids [B, K], probabilities [B, K] as a CPU tensor, retrieved
    count [B], filter survivors [B]) with K = the largest retrieved count of the batch; slots behind a question's
    count hold (pad entity, 0.0)."""
    local_entity = np.asarray(local_entity)
    # evaluate.py:177,198-205: the seed flags are compared after a cast to int64
    eligible = (np.asarray(query_entities).astype(np.int64) != 1) & (local_entity != pad_ent_id)
    el = torch.from_numpy(eligible.astype(np.uint8)).to(pred_dist.device)
    pred_dist = pred_dist.detach().float().contiguous()
    slots, cnt = ops.topp_candidates(pred_dist, el, ignore_prob, eps)
    cnt_h = cnt.cpu().numpy()
    K = max(1, int(cnt_h[:, 1].max()) if len(cnt_h) else 1)
    head = slots[:, :K].long().clamp_(min=0)
    live = torch.arange(K, device=pred_dist.device)[None, :] < cnt[:, 1:2].long()
    probs = torch.where(live, torch.gather(pred_dist, 1, head), torch.zeros((), device=pred_dist.device)).cpu()
    head_h, live_h = head.cpu().numpy(), (np.arange(K)[None, :] < cnt_h[:, 1:2])
    ents = np.where(live_h, np.take_along_axis(local_entity, head_h, axis=1), pad_ent_id).astype(local_entity.dtype)
    return ents, probs, cnt_h[:, 1].astype(np.int64), cnt_h[:, 0].astype(np.int64)


def retrieved_candidates(pred_dist: torch.Tensor, local_entity: np.ndarray, query_entities: np.ndarray,
                         pad_ent_id: int, ignore_prob: float, eps: float):
    """Per question the list ``[(entity id, prob), ...]`` that ``f1_and_hits`` would retrieve, best first,
    and the number of candidates that passed the filter.  ``pred_dist`` stays on the GPU."""
    ents, probs, k, passed = _retrieved_arrays(pred_dist, local_entity, query_entities, pad_ent_id, ignore_prob, eps)
    probs = probs.numpy()
    return [(list(zip(ents[b, :k[b]].tolist(), probs[b, :k[b]].tolist())), int(passed[b])) for b in range(len(k))]


def compact_batch(pred_dist: torch.Tensor, local_entity: np.ndarray, query_entities: np.ndarray, pad_ent_id: int,
                  ignore_prob: float, eps: float):
    """(local_entity', query_entities', pred_dist') of width K = the largest retrieved count of the batch: per question
    the retrieved slots best first, then pad entities with probability 0 (the reference's filter skips them,
    evaluate.py:201-202).  pred_dist' is a CPU tensor (the reference calls ``.tolist()`` on its rows).  No per-question
    Python: one gather on the device, one ``take_along_axis`` on the host."""
    ents, probs, k, _ = _retrieved_arrays(pred_dist, local_entity, query_entities, pad_ent_id, ignore_prob, eps)
    STATS["questions"] += int(len(k))
    STATS["retrieved"] += int(k.sum())
    STATS["retrieved_max"] = max(STATS["retrieved_max"], int(k.max()) if len(k) else 0)
    return ents, np.zeros(ents.shape, dtype=np.asarray(query_entities).dtype), probs


class _CompactingModel:
    """Stands in for ``Evaluator.model`` during one ``evaluate`` call: runs the real forward, then swaps the batch's
    candidate arrays and the returned distribution for their compacted forms."""

    def __init__(self, model, state, pad_ent_id, ignore_prob, eps):
        self.__dict__.update(_m=model, _s=state, _a=(pad_ent_id, ignore_prob, eps))

    def __getattr__(self, name):
        return getattr(self._m, name)

    def __call__(self, batch, *args, **kw):
        loss, extras, pred_dist, tp_list = self._m(batch, *args, **kw)
        full = self._s["batch"]
        if pred_dist.shape[1] <= TOPP_MAX_N:
            full[0], full[1], pred_dist = compact_batch(pred_dist, full[0], full[1], *self._a)
        return loss, extras, pred_dist, tp_list


def _delegating(orig_evaluate):
    def evaluate(self, valid_data, test_batch_size=20, write_info=False):
        """``Evaluator.evaluate`` (evaluate.py:147-240) itself, on device-compacted batches."""
        import torch.distributed as dist
        state = {}
        get_batch, model = valid_data.get_batch, self.model

        def get_batch_list(*a, **kw):
            state["batch"] = list(get_batch(*a, **kw))      # evaluate unpacks the batch AFTER the forward (:165-170)
            return state["batch"]

        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0 and self.file_write is None:
            self.file_write = open(os.devnull, "w")         # question-sharded run: every rank scores, rank 0 reports
        valid_data.get_batch = get_batch_list
        self.model = _CompactingModel(model, state, len(self.id2entity), (1 - self.eps) / valid_data.max_local_entity, self.eps)
        try:
            return orig_evaluate(self, valid_data, test_batch_size, write_info)
        finally:
            self.model = model
            del valid_data.get_batch                        # the instance attribute that shadowed the bound method
    return evaluate


def patch_evaluator_class(evaluator_cls):
    """``Evaluator.evaluate`` of the reference's class -> the delegating form (tools/run_reference.py)."""
    if not getattr(evaluator_cls.evaluate, "_gnnrag_delegating", False):
        fn = _delegating(evaluator_cls.evaluate)
        fn._gnnrag_delegating = True
        evaluator_cls.evaluate = fn
    return evaluator_cls


def patch_evaluator(evaluator):
    """Rebinds ``evaluator.evaluate`` (a reference ``Evaluator`` instance) to the delegating form."""
    evaluator.evaluate = types.MethodType(_delegating(type(evaluator).evaluate), evaluator)
    return evaluator
