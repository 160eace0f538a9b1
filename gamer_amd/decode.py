"""Evaluation path of the SMB decoder: trie-constrained beam search over the HIP engine.

Mirrors what ``test_SMB_decoder.py:test_single_behavior`` obtains from
``model.generate(..., max_new_tokens=4, num_beams=k, num_return_sequences=k, prefix_allowed_tokens_fn=...,
early_stopping=True)`` (ref:SeqRec/tasks/test_SMB_decoder.py:163-180) with the per-behaviour ``Trie`` of
``prefix_allowed_tokens_fn_by_last_token`` (ref:SeqRec/generation/trie.py:5-104; built at
test_SMB_decoder.py:470-500): the same sequences in the same order and the same ``sequences_scores``.

* The trie lives on the device as a CSR array; ``gamer_trie_logprobs`` does log-softmax + constraint + beam
  score per row and ``gamer_trie_advance`` moves the beams' trie nodes (csrc/decode.hip).  The reference walks a
  Python dict per (sample, beam) per step on the host.
* Beam bookkeeping follows transformers' ``GenerationMixin._beam_search`` for this call: top 2k of k*V
  candidates, the best k continue, the best k of the last step are the hypotheses, score = sum of log-probs /
  number of new tokens.  No EOS can be produced (the trie never allows it).
* No K/V cache yet: the engine re-runs the whole sequence each step (4 steps).  The two things the reference's
  cache freezes are passed explicitly (``act_zero_col``, ``uniform_len``; Engine.forward).  One reference defect
  is NOT reproduced: its cross-attention cache lives on the module and is not re-ordered with the beams
  (model.py:569,785,844-860), which perturbs samples whose target row is "empty" (oracle/decode_oracle.py,
  tests/test_decode.py quantify it).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch

from . import ops


class ItemTrie:
    """CSR trie over item token sequences (node 0 = root).  ``get(prefix)`` is the host-side ``Trie.get``."""

    def __init__(self, sequences: Sequence[Sequence[int]], device="cuda"):
        nodes: List[Dict[int, int]] = [{}]
        for seq in sequences:
            cur = 0
            for t in seq:
                t = int(t)
                nxt = nodes[cur].get(t)
                if nxt is None:
                    nxt = len(nodes)
                    nodes[cur][t] = nxt
                    nodes.append({})
                cur = nxt
        start, tok, child = [0], [], []
        for d in nodes:
            for t, c in d.items():             # insertion order, as the reference's dict
                tok.append(t)
                child.append(c)
            start.append(len(tok))
        self.nodes = nodes
        self.n_items = len(sequences)
        # tokens that end an item, plus the pad id (test_SMB_decoder.py:473-474)
        self.last_tokens = {int(seq[-1]) for seq in sequences} | {4}
        i32 = dict(dtype=torch.int32, device=device)
        self.child_start = torch.tensor(start, **i32)
        self.child_tok = torch.tensor(tok if tok else [0], **i32)
        self.child_node = torch.tensor(child if child else [0], **i32)

    def get(self, prefix: Sequence[int]) -> List[int]:
        cur = 0
        for t in prefix:
            cur = self.nodes[cur].get(int(t))
            if cur is None:
                return []
        return list(self.nodes[cur].keys())


def prefix_allowed_tokens(trie: ItemTrie):
    """Callable with the signature of the reference's ``prefix_allowed_tokens_fn_by_last_token`` result
    (trie.py:90-104) for host-side use; ``Qwen3MultiWithTemperature.generate`` takes the trie it carries."""
    last_tokens = {seq_last for seq_last in trie.last_tokens}

    def fn(batch_id: int, sentence) -> List[int]:
        s = [int(t) for t in sentence]
        i = len(s) - 1
        while i >= 0 and s[i] not in last_tokens:
            i -= 1
        return trie.get(s[i + 1:])
    fn.trie = trie
    return fn


@torch.no_grad()
def beam_search(engine, input_ids: torch.Tensor, attention_mask: torch.Tensor, actions: torch.Tensor, trie: ItemTrie,
                num_beams: int, max_new_tokens: int = 4) -> Tuple[torch.Tensor, torch.Tensor]:
    """input_ids / attention_mask / actions: [B, L0] left-padded prompts ending with the target behaviour token.
    Returns (sequences [B*num_beams, L0+max_new_tokens] int64, sequences_scores [B*num_beams] fp32), the beams of
    sample b at rows b*num_beams .., best first - the layout of HF's GenerateBeamOutput."""
    dev = engine.device
    B, L0 = input_ids.shape
    V = engine.cfg.vocab_size
    nb, K = num_beams, 2 * num_beams
    N = B * nb
    ids0 = input_ids.to(dev, torch.int64)
    am0 = attention_mask.to(dev, torch.int64)
    act0 = actions.to(dev, torch.int64)
    seqs = ids0[:, None, :].expand(B, nb, L0).contiguous()
    run_scores = torch.zeros(B, nb, device=dev)
    run_scores[:, 1:] = -1e9                       # only beam 0 is live at the first step (HF)
    # trie node of every beam: the prefix since the last complete item is the target behaviour token
    node = torch.zeros(N, dtype=torch.int32, device=dev)
    nxt = torch.empty_like(node)
    ops.trie_advance(node, ids0[:, -1].repeat_interleave(nb).contiguous(), trie.child_start, trie.child_tok,
                     trie.child_node, nxt)
    node, nxt = nxt, node
    scores = torch.empty(N, V, device=dev)
    final = None
    for step in range(max_new_tokens):
        cur = L0 + step
        if step == 0:
            # all beams of a sample hold the same prompt: run it once (HF runs num_beams copies)
            engine.forward(ids0, am0, act0, train=False, act_zero_col=L0 - 1, uniform_len=L0)
            rows = (torch.arange(N, device=dev, dtype=torch.int32) // nb) * cur + (cur - 1)
        else:
            flat = seqs.reshape(N, cur)
            am = torch.cat([am0, am0.new_ones(B, step)], 1).repeat_interleave(nb, 0)
            # generated tokens belong to the target item: same behaviour level as its behaviour token
            act = torch.cat([act0, act0[:, -1:].expand(B, step)], 1).repeat_interleave(nb, 0)
            engine.forward(flat, am, act, train=False, act_zero_col=L0 - 1, uniform_len=L0)
            rows = torch.arange(N, device=dev, dtype=torch.int32) * cur + (cur - 1)
        ws = engine.ws
        ops.trie_logprobs(ws.logits, rows, run_scores.reshape(N).contiguous(), node, trie.child_start, trie.child_tok,
                          V, scores)
        top_s, top_i = torch.topk(scores.view(B, nb * V), K)
        beam_i, tok = top_i // V, top_i % V
        cand = torch.cat([torch.gather(seqs, 1, beam_i[:, :, None].expand(B, K, cur)), tok[:, :, None]], 2)
        if step == max_new_tokens - 1:
            final = (cand[:, :nb].reshape(N, cur + 1), (top_s[:, :nb] / max_new_tokens).reshape(N))
            break
        seqs = cand[:, :nb].contiguous()
        run_scores = top_s[:, :nb].contiguous()
        parent = (beam_i[:, :nb] + torch.arange(B, device=dev)[:, None] * nb).reshape(N)
        ops.trie_advance(node[parent].contiguous(), tok[:, :nb].reshape(N).contiguous(), trie.child_start,
                         trie.child_tok, trie.child_node, nxt)
        node, nxt = nxt.clone(), nxt
    return final
