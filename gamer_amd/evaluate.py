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
                                   num_beams, item_len, session_ids=batch.get("session_ids"),
                                   extended_session_ids=batch.get("extended_session_ids"))
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


@torch.no_grad()
def evaluate_dataset(engine, data, max_his_len: int, behaviors: Sequence[str] = None, num_beams: int = 20,
                     batch_size: int = 64, metric_list: Sequence[str] = ("hit@1", "hit@5", "hit@10", "ndcg@5", "ndcg@10"),
                     rank: int = 0, world: int = 1) -> Dict[str, Dict[str, float]]:
    """``TestSMBDecoder.test`` over a dataset directory (test_SMB_decoder.py:408-440, 455-540): for every behaviour
    the test users that hold it in their last session, prompts from ``gamer_amd.data.Collator.test`` (history + the
    behaviour token), beams constrained to that behaviour's item trie.  ``data`` is a ``gamer_amd.data.SMBData``."""
    from . import data as gdata
    samples = data.test_samples(max_his_len)
    coll = gdata.Collator(data)
    out = {}
    for beh in (behaviors if behaviors is not None else data.behaviors):
        sub = samples.filter_by_behavior(beh)
        trie = ItemTrie(data.candidate_tokens(beh).tolist(), device=engine.device,
                        pad_token_id=engine.cfg.pad_token_id)

        def gen():
            for idx in gdata.batches(len(sub), batch_size, rank=rank, world=world):
                inputs, targets = coll.test(sub, idx, behavior=beh)
                yield {"input_ids": inputs["input_ids"], "attention_mask": inputs["attention_mask"],
                       "actions": inputs["actions"], "session_ids": inputs["session_ids"],
                       "extended_session_ids": inputs["extended_session_ids"],
                       "targets": [t.tolist() for t in targets]}
        out[beh] = evaluate_behavior(engine, gen(), trie, num_beams, metric_list, item_len=data.sole_item_len)
    return out
