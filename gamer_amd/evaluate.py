"""Evaluation loop of one target behaviour: beam search + ranking metrics.

Mirror of ``TestSMBDecoder.test_single_behavior`` (ref:SeqRec/tasks/test_SMB_decoder.py:90-285) for the
Qwen3Multi backbone: every batch is a dict with left-padded ``input_ids`` / ``attention_mask`` / ``actions``
that already end with the target behaviour token (test_SMB_decoder.py:112-118) and ``targets``: per sample the
list of held-out items (token tuples of the 4 semantic IDs).  Under ``torch.distributed`` every rank evaluates its
own batches and the metric sums are all-reduced (the reference gathers Python objects; the sums are the same).
"""
from __future__ import annotations

from typing import Dict, Iterable, Sequence

import torch

from . import metrics as gm
from .decode import ItemTrie, beam_search


@torch.no_grad()
def evaluate_behavior(engine, batches: Iterable[Dict], trie: ItemTrie, num_beams: int = 20,
                      metric_list: Sequence[str] = ("hit@1", "hit@5", "hit@10", "ndcg@5", "ndcg@10"),
                      item_len: int = 4) -> Dict[str, float]:
    sums = {m: 0.0 for m in metric_list}
    total = 0
    for batch in batches:
        seqs, scores = beam_search(engine, batch["input_ids"], batch["attention_mask"], batch["actions"], trie,
                                   num_beams, item_len)
        pred = seqs[:, -item_len:].cpu().tolist()
        topk = gm.get_topk_results(pred, scores.cpu().tolist(), batch["targets"], num_beams)
        res = gm.get_metrics_results(topk, metric_list, batch["targets"])
        for m in metric_list:
            sums[m] += res[m]
        total += len(batch["targets"])
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        vec = torch.tensor([sums[m] for m in metric_list] + [float(total)], dtype=torch.float64,
                           device=engine.device if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(vec)
        for i, m in enumerate(metric_list):
            sums[m] = float(vec[i])
        total = int(vec[-1])
    out = {m: sums[m] / max(total, 1) for m in metric_list}
    out["samples"] = total
    return out
