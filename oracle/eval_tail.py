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


def f1_and_hits(answers, candidate2prob, id2entity, entity2name, eps):
    """``f1_and_hits`` of the reference (``gnn/evaluate.py:24-67``) restated: candidates sorted by probability
    (stable), retrieved until the running sum exceeds eps; returns (precision, recall, f1, hit, em, case,
    retrieved [(name, prob)], answer names).  Pinned by tests/test_oracle_golden.py against the ``.info`` lines the
    live reference wrote (tests/golden/rearev_closed_loop.npz)."""
    name = (lambda c: id2entity[c]) if entity2name is None else (lambda c: entity2name[id2entity[c]])
    ans = [name(a) for a in answers]
    retrieved, correct = [], 0
    cand_list = sorted(candidate2prob, key=lambda x: x[1], reverse=True)      # evaluate.py:34
    best_ans = cand_list[0][0] if cand_list else -1                            # :35-38
    tp_prob = 0.0
    for c, prob in cand_list:                                                  # :41-50
        retrieved.append((name(c), prob))
        tp_prob += prob
        if c in answers:
            correct += 1
        if tp_prob > eps:
            break
    em = 1 if correct > 0 else 0
    if len(answers) == 0:                                                      # :55-59
        if len(retrieved) == 0:
            return 1.0, 1.0, 1.0, 1.0, 1.0, 0, retrieved, ans
        return 0.0, 1.0, 0.0, 1.0, 1.0, 1, retrieved, ans
    hits = float(best_ans in answers)                                          # :61
    if len(retrieved) == 0:
        return 1.0, 0.0, 0.0, hits, hits, 2, retrieved, ans
    p, r = correct / len(retrieved), correct / len(answers)
    f1 = 2.0 / (1.0 / p + 1.0 / r) if p != 0 and r != 0 else 0.0
    return p, r, f1, hits, em, 3, retrieved, ans


def info_record(question, num_iter, answers, candidate2prob, id2entity, entity2name, eps):
    """One line of ``<experiment>_test.info`` as ``Evaluator.evaluate`` writes it in eval mode (tp_list is None:
    ``write_info`` leaves an empty dict per iteration, evaluate.py:120-144; then :210-219)."""
    precision, recall, f1, hit, em, _, retrieved, ans = f1_and_hits(answers, candidate2prob, id2entity, entity2name, eps)
    obj = {"question": question}
    for j in range(num_iter):
        obj[j] = {}
    obj.update(answers=ans, precison=precision, recall=recall, f1=f1, hit=hit, em=em, cand=retrieved)
    return obj
