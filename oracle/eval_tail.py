"""ORACLE (test infrastructure, NOT product code) - plain-Python restatement of the candidate selection
in ``Evaluator.evaluate`` (reference ``gnn/evaluate.py:188-207``) followed by the sort and top-p cut of
``f1_and_hits`` (``evaluate.py:34-51``).  Only ``tests/`` may import this file.

Pinning: ``tests/test_eval_tail.py`` runs the live reference's ``Evaluator.evaluate`` next to the
patched one (identical metrics and an identical ``.info`` file)."""
from __future__ import annotations


def select(probs, candidates, seeds, pad_ent_id, ignore_prob, eps):
    """One question.  probs: list of float (``pred_dist[b].tolist()``), candidates: entity ids,
    seeds: ``query_entities[b].tolist()``.  Returns (kept slots sorted, number retrieved)."""
    kept = []
    for j, (c, p, s) in enumerate(zip(candidates, probs, seeds)):
        if s == 1.0:                      # evaluate.py:198-203: seeds are never answers
            continue
        if c == pad_ent_id:               # :204-205
            continue
        if p < ignore_prob:               # :206-207
            continue
        kept.append((j, p))
    kept = sorted(kept, key=lambda x: x[1], reverse=True)       # evaluate.py:34 (stable)
    tp_prob, cut = 0.0, 0
    for _, p in kept:                                            # evaluate.py:41-50
        tp_prob += p
        cut += 1
        if tp_prob > eps:
            break
    return [j for j, _ in kept], cut
