"""Ranking metrics of the evaluation path, on token-id tuples.

Mirror of ref:SeqRec/evaluation/ranking.py:5-90 (``get_topk_results``, ``hit_k``, ``ndcg_k``, ``recall_k``,
``get_metrics_results``).  The reference compares decoded strings; item strings and their token tuples are in
1:1 correspondence, so tuples are compared directly.  Targets are lists of items per sample (the SMB test sets
hold every held-out item of the target behaviour), as at test_SMB_decoder.py:101.
Values are SUMS over the samples, as upstream; divide by the sample count at the end (test_SMB_decoder.py:285).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence


def get_topk_results(pred_items, scores, targets: Sequence[Sequence[Sequence[int]]], k: int) -> List[List[int]]:
    """pred_items: [B*k] token sequences, scores: [B*k], targets[b]: list of target items of sample b.
    Returns per sample the 0/1 hit list of its k predictions ordered by score (descending, stable)."""
    out = []
    for b in range(len(targets)):
        rows = [(tuple(int(t) for t in pred_items[b * k + j]), float(scores[b * k + j])) for j in range(k)]
        rows.sort(key=lambda x: x[1], reverse=True)
        tgt = {tuple(int(t) for t in item) for item in targets[b]}
        out.append([1 if r[0] in tgt else 0 for r in rows])
    return out


def hit_k(topk: List[List[int]], k: int) -> float:
    return float(sum(1 for row in topk if sum(row[:k]) > 0))


def recall_k(topk: List[List[int]], k: int, n_targets: Sequence[int]) -> float:
    return float(sum(min(sum(row[:k]), n) / n for row, n in zip(topk, n_targets)))


def ndcg_k(topk: List[List[int]], k: int, n_targets: Sequence[int]) -> float:
    total = 0.0
    for row, n in zip(topk, n_targets):
        dcg, cnt = 0.0, 0
        for j, r in enumerate(row[:k]):
            if r == 1:
                cnt += 1
            dcg += r / math.log(j + 2, 2)
            if cnt == n:
                break
        ideal = sum(1.0 / math.log(j + 2, 2) for j in range(min(k, n)))
        total += dcg / ideal
    return total


def get_metrics_results(topk: List[List[int]], metrics: Sequence[str],
                        targets: Sequence[Sequence[Sequence[int]]]) -> Dict[str, float]:
    n_targets = [len({tuple(int(t) for t in item) for item in tg}) for tg in targets]
    res = {}
    for m in metrics:
        kind, k = m.lower().split("@")
        k = int(k)
        if kind == "hit":
            res[m] = hit_k(topk, k)
        elif kind == "ndcg":
            res[m] = ndcg_k(topk, k, n_targets)
        elif kind == "recall":
            res[m] = recall_k(topk, k, n_targets)
        else:
            raise NotImplementedError(m)
    return res
