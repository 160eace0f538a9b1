"""CPU restatement of the evaluation path (TEST INFRASTRUCTURE ONLY - never imported by gamer_amd/).

* ``ItemTrie``: ref:SeqRec/generation/trie.py:5-104 with ``prefix_allowed_tokens_fn_by_last_token``
  (:90-104) as test_SMB_decoder.py:470-500 builds it per behaviour: the allowed next tokens are the children of
  the trie node reached by the tokens generated since the prompt's last item.
* ``beam_search``: transformers ``GenerationMixin._beam_search`` (third party, pinned 4.51.0 upstream, 5.15.0
  executed for the fixture) for the call of test_SMB_decoder.py:163-180: ``num_beams`` beams, ``max_new_tokens``
  steps, no EOS can be produced (the trie never allows it), ``length_penalty`` 1, ``early_stopping`` True:
      log_probs = log_softmax(logits[:, -1]);  disallowed -> -inf;  + running score;
      top 2*num_beams over beams x vocab;  running beams = best num_beams;  at the last step the best num_beams
      candidates are the finished hypotheses with score = sum_logprob / max_new_tokens.
  The model is re-run on the whole sequence every step (no cache), with the two things a cache freezes passed
  explicitly: the prompt's last token keeps action index 0 (``act_zero_col``) and an "empty" row stays uniform
  over the keys that existed when it was computed (``uniform_len``).  That is equivalent to the reference's cached
  decode (model.py:573-630, 691-741; router.py:83-97) up to ONE thing the reference's cache does that a re-run
  cannot see by itself: the reference keeps the cross-attention K/V cache on the module (model.py:569, 785,
  844-860) and HF only reorders ``past_key_values``, so after a beam re-ordering the cross cache rows of GENERATED
  positions belong to whichever beam sat in that SLOT when they were written.  Those keys are masked for every query
  except "empty" rows (no lower-level key in the prompt), whose uniform attention then averages V rows of other
  beams.  ``reorder_cross_cache=False`` (the default: the reference as shipped) restates exactly that with a
  per-slot, append-only store of the generated positions' cross V rows and the slot every beam occupied when each of
  its positions was computed; ``True`` is the variant with the cross cache re-ordered like the self cache (what the
  fixture calls ``_crossfix``).  Rows with a non-empty mask are identical in both.
* Qwen3SessionMulti (``session_ids`` / ``extended_session_ids`` given): its cached steps use a plain causal self
  mask and the prompt's last cross-mask row (Qwen3SessionMulti/model.py:598-613, 716-728) and advance the RoPE
  position by one per generated token from the prompt's largest extended id (:969-982).  Re-running the whole
  sequence with the TRAINING masks gives exactly that when the generated tokens carry the behaviour token's
  session id (they sit in its item block and every history key belongs to an earlier session, so the self mask
  lets them see everything before them; their own session is excluded from the cross mask like the generated keys
  are in the cached row) and extended ids last + 1, last + 2, ...
* ``ranking``: ref:SeqRec/evaluation/ranking.py:5-90 on token tuples instead of decoded strings.

Pinned by tests/golden/decode_small.npz (oracle/make_golden_decode.py runs the real reference).
"""
import math
from typing import Dict, List, Sequence

import torch

from . import qwen3multi_oracle as orc


class ItemTrie:
    def __init__(self, sequences: Sequence[Sequence[int]]):
        self.root: Dict[int, dict] = {}
        for seq in sequences:
            d = self.root
            for t in seq:
                d = d.setdefault(int(t), {})

    def get(self, prefix: Sequence[int]) -> List[int]:
        d = self.root
        for t in prefix:
            if int(t) not in d:
                return []
            d = d[int(t)]
        return list(d.keys())


class _SlotCrossCache:
    """The generated part of the reference's module-level cross-attention cache under beam search: row (slot n,
    generated index g) is written once, by the beam that sits in slot n when position L0 + g is computed, and never
    moves.  ``hist[n][g]`` = the slot the beam NOW in slot n occupied at that time."""

    def __init__(self, N: int, L0: int):
        self.N, self.L0 = N, L0
        self.v: Dict[str, Dict[int, torch.Tensor]] = {}          # layer prefix -> generated index -> [N, nkv, dh]
        self.hist: List[List[int]] = [[] for _ in range(N)]

    def hook(self, step: int, nq: int):
        """o_hook of orc.attention for the forward of generation step ``step`` (>= 1: positions L0 .. L0+step-1 exist)."""
        L0 = self.L0

        def fn(prefix, o, v, empty):
            store = self.v.setdefault(prefix, {})
            g_new = step - 1
            store[g_new] = v[:, L0 + g_new].detach().clone()      # this forward's newest position, written at its slot
            rep = nq // v.shape[2]
            o = o.clone()
            for n in range(self.N):
                slots = self.hist[n] + [n]                        # slot at positions g = 0 .. g_new
                for g in range(step):
                    if not bool(empty[n, L0 + g]):
                        continue
                    acc = v[n, :L0].sum(0)
                    for g2 in range(g + 1):
                        acc = acc + store[g2][slots[g]]
                    o[n, L0 + g] = (acc / float(L0 + g + 1)).repeat_interleave(rep, 0)
            return o
        return fn

    def advance(self, parent: List[int]):
        """after the beams were re-ordered: new slot n continues the beam that sat in slot parent[n]"""
        self.hist = [self.hist[p] + [p] for p in parent]


def beam_search(sd, cfg, input_ids, attention_mask, actions, trie: ItemTrie, num_beams: int, max_new_tokens: int = 4,
                forward=None, session_ids=None, extended_session_ids=None, reorder_cross_cache: bool = False):
    """Returns (sequences [B*num_beams, L0+max_new_tokens], sequences_scores [B*num_beams]), best beam first.
    ``reorder_cross_cache``: see the module docstring (False = the reference as shipped)."""
    B, L0 = input_ids.shape
    V = sd["model.embed_tokens.weight"].shape[0]
    K = 2 * num_beams
    slot_cache = _SlotCrossCache(B * num_beams, L0) if (not reorder_cross_cache and forward is None) else None
    seqs = input_ids[:, None, :].expand(B, num_beams, L0).clone()
    run_scores = torch.zeros(B, num_beams)
    run_scores[:, 1:] = -1e9
    am0, act0 = attention_mask, actions
    fin_seq, fin_score = None, None
    for step in range(max_new_tokens):
        cur = L0 + step
        flat = seqs.reshape(B * num_beams, cur)
        am = torch.cat([am0, torch.ones(B, step, dtype=am0.dtype)], 1).repeat_interleave(num_beams, 0)
        # generated tokens belong to the target item: same behaviour level as its behaviour token
        act = torch.cat([act0, act0[:, -1:].expand(B, step)], 1).repeat_interleave(num_beams, 0)
        skw = {}
        if session_ids is not None:
            sess = torch.cat([session_ids, session_ids[:, -1:].expand(B, step)], 1)
            ext = torch.cat([extended_session_ids, extended_session_ids[:, -1:] + torch.arange(1, step + 1)[None, :]], 1)
            skw = dict(session_ids=sess.repeat_interleave(num_beams, 0),
                       extended_session_ids=ext.repeat_interleave(num_beams, 0))
        if forward is None:
            hook = slot_cache.hook(step, cfg.num_attention_heads) if (slot_cache is not None and step >= 1) else None
            with torch.no_grad():
                logits = orc.forward(sd, cfg, flat, am, act, act_zero_col=L0 - 1, uniform_len=L0, cross_o_hook=hook,
                                     **skw)["logits"][:, -1].float()
        else:
            logits = forward(flat, am, act, L0 - 1, **skw)
        logp = torch.log_softmax(logits, -1)
        masked = torch.full_like(logp, -math.inf)
        for n in range(B * num_beams):
            allowed = trie.get(flat[n, L0 - 1:].tolist())
            if allowed:
                masked[n, allowed] = logp[n, allowed]
        cand = (masked.view(B, num_beams, V) + run_scores[:, :, None]).view(B, num_beams * V)
        top_s, top_i = torch.topk(cand, K)
        beam_i, tok = top_i // V, top_i % V
        cand_seq = torch.cat([torch.gather(seqs, 1, beam_i[:, :, None].expand(B, K, cur)), tok[:, :, None]], 2)
        if step == max_new_tokens - 1:
            fin_seq, fin_score = cand_seq[:, :num_beams], top_s[:, :num_beams] / max_new_tokens
        seqs, run_scores = cand_seq[:, :num_beams].clone(), top_s[:, :num_beams].clone()
        if slot_cache is not None and step >= 1:
            parent = (beam_i[:, :num_beams] + torch.arange(B)[:, None] * num_beams).reshape(-1).tolist()
            slot_cache.advance(parent)
    return fin_seq.reshape(B * num_beams, -1), fin_score.reshape(-1)


# ---- ranking.py ------------------------------------------------------------------------------
def topk_results(pred_items, scores, targets, k: int):
    """pred_items [B*k, n] token ids, scores [B*k], targets [B, n] -> per sample the hit list ordered by score."""
    out = []
    for b in range(len(targets)):
        rows = [(tuple(int(t) for t in pred_items[b * k + j]), float(scores[b * k + j])) for j in range(k)]
        rows.sort(key=lambda x: x[1], reverse=True)
        tgt = tuple(int(t) for t in targets[b])
        out.append([1 if r[0] == tgt else 0 for r in rows])
    return out


def metrics(topk, names):
    """Sums over the batch (the reference divides by the sample count at the end), one target per sample."""
    res = {}
    for m in names:
        kind, k = m.lower().split("@")
        k = int(k)
        if kind == "hit":
            res[m] = float(sum(1 for row in topk if sum(row[:k]) > 0))
        elif kind == "recall":
            res[m] = float(sum(min(sum(row[:k]), 1) for row in topk))
        elif kind == "ndcg":
            tot = 0.0
            for row in topk:
                for j, r in enumerate(row[:k]):
                    if r == 1:
                        tot += 1.0 / math.log(j + 2, 2)
                        break
            res[m] = tot
        else:
            raise NotImplementedError(m)
    return res
