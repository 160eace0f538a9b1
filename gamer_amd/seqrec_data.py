"""`train_decoder` plumbing of BASELINE configs[0] (TIGER single-behaviour decoder on Beauty, CPU): the reference's
``SeqRecDataset`` + ``T5Tokenizer`` + ``EncoderDecoderCollator`` as a pre-tokeniser that emits id tensors directly, and
TIGER's temperature cross-entropy through the HIP loss kernels.  SURVEY.md section 8(f) row 4 / section 3.4; no T5 kernels
(the T5 encoder-decoder itself is outside the MI355X path, docs/DESIGN_rounds1-4.md section 14).

Reference behaviour restated (bit-exact against ``tests/golden/seqrec_small.npz``, generated from the real classes by
``oracle/make_golden_seqrec.py``):
  on-disk format   ref:SeqRec/datasets/seq_dataset.py:81-87  (<name>.inter.json {user: [item, ...]},
                   <name>.index.json {item: [token, ...]})
  samples          seq_dataset.py:95-143: leave-last-two-out - train = every prefix of items[:-2] with >= 1 history item,
                   valid = items[-2] given items[:-2], test = items[-1] given items[:-1]; history cropped to the last
                   ``max_his_len`` items when max_his_len > 0
  vocabulary       seq_dataset.py:26-36 + ref:SeqRec/tasks/train_decoder.py:239-240: ``T5Tokenizer`` of
                   ref:config/s2s-models/TIGER (32,100 entries) + ``add_tokens(sorted(new_tokens))`` -> ids 32100, ...
  collation        ref:SeqRec/datasets/collator.py:7-44: text and text_target tokenised with an appended </s> (id 1),
                   truncated to ``model_max_length`` (the first max - 1 tokens + </s>), right-padded with 0 to the longest
                   row of the batch; labels == pad -> -100
  loss             ref:SeqRec/models/generative/TIGER/model.py:153-161: CrossEntropyLoss(ignore_index=-100) of
                   lm_logits / temperature against the UNshifted labels (mean over the labelled positions)
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

T5_BASE_VOCAB = 32100       # len(T5Tokenizer.from_pretrained(ref:config/s2s-models/TIGER)): 32,000 pieces + 100 <extra_id_*>
T5_PAD_ID = 0               # ref:config/s2s-models/TIGER/config.json pad_token_id / decoder_start_token_id
T5_EOS_ID = 1               # eos_token_id
IGNORE_INDEX = -100


@dataclass
class SeqSamples:
    """Samples in CSR form: sample n's history is ``hist[ptr[n]:ptr[n+1]]`` (item rows), its target ``target[n]``."""
    mode: str
    ptr: np.ndarray
    hist: np.ndarray
    target: np.ndarray

    def __len__(self) -> int:
        return len(self.target)


class SeqRecData:
    """``SeqRecDataset`` (seq_dataset.py:54-144) on id arrays: items become rows of token ids once."""

    def __init__(self, data_path: str, dataset: str, index_file: str = ".index.json", inter_type: Optional[str] = None,
                 base_vocab_size: int = T5_BASE_VOCAB):
        d = os.path.join(data_path, dataset)
        suffix = f"{inter_type}.inter" if inter_type is not None else "inter"
        with open(os.path.join(d, f"{dataset}.{suffix}.json")) as f:
            self.inters: Dict[str, List[int]] = json.load(f)
        with open(os.path.join(d, dataset + index_file)) as f:
            indices: Dict[str, List[str]] = json.load(f)
        # get_new_tokens(): the sorted set of every index token; add_tokens() hands out ids in that order
        self.new_tokens: List[str] = sorted({t for idx in indices.values() for t in idx})
        self.token_id = {t: base_vocab_size + i for i, t in enumerate(self.new_tokens)}
        self.vocab_size = base_vocab_size + len(self.new_tokens)
        # item -> row of token ids (CSR: items need not all have the same number of tokens)
        self.item_row = {int(k): n for n, k in enumerate(indices)}
        lens = np.array([len(v) for v in indices.values()], dtype=np.int64)
        self.item_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        self.item_tok = np.array([self.token_id[t] for v in indices.values() for t in v], dtype=np.int64)
        self.item_str = {int(k): "".join(v) for k, v in indices.items()}

    def all_items(self):
        """get_all_items(): (distinct item strings, strings that occur for more than one item)."""
        seen, coll = set(), set()
        for s in self.item_str.values():
            (coll if s in seen else seen).add(s)
        return seen, coll

    def samples(self, mode: str, max_his_len: int) -> SeqSamples:
        ptr, hist, target = [0], [], []

        def add(history: Sequence[int], tgt: int):
            if max_his_len > 0:
                history = history[-max_his_len:]
            hist.extend(self.item_row[i] for i in history)
            ptr.append(len(hist))
            target.append(self.item_row[tgt])

        for items in self.inters.values():
            if mode == "train":
                body = items[:-2]
                for i in range(1, len(body)):
                    add(body[:i], body[i])
            elif mode == "valid":
                add(items[:-2], items[-2])
            elif mode == "test":
                add(items[:-1], items[-1])
            else:
                raise NotImplementedError(mode)
        return SeqSamples(mode, np.array(ptr, np.int64), np.array(hist, np.int64), np.array(target, np.int64))


class EncoderDecoderCollator:
    """collator.py:7-44 without strings: ``input_ids`` / ``attention_mask`` [B, Lin], ``labels`` [B, Lout], ``split``."""

    def __init__(self, data: SeqRecData, model_max_length: int = 512):
        self.data = data
        self.max_len = int(model_max_length)

    def _rows(self, item_lists: List[np.ndarray]) -> List[np.ndarray]:
        d = self.data
        out = []
        for items in item_lists:
            if len(items):
                toks = np.concatenate([d.item_tok[d.item_ptr[i]:d.item_ptr[i + 1]] for i in items])
            else:
                toks = np.zeros(0, np.int64)
            # truncation=True, max_length=model_max_length: the tokenizer keeps the first max - 1 tokens and appends </s>
            out.append(np.concatenate([toks[:self.max_len - 1], [T5_EOS_ID]]).astype(np.int64))
        return out

    @staticmethod
    def _pad(rows: List[np.ndarray], fill: int) -> np.ndarray:
        L = max(len(r) for r in rows)
        out = np.full((len(rows), L), fill, dtype=np.int64)
        for n, r in enumerate(rows):
            out[n, :len(r)] = r
        return out

    def __call__(self, samples: SeqSamples, index: Sequence[int]) -> Dict[str, torch.Tensor]:
        ins = self._rows([samples.hist[samples.ptr[n]:samples.ptr[n + 1]] for n in index])
        outs = self._rows([samples.target[n:n + 1] for n in index])
        ids = self._pad(ins, T5_PAD_ID)
        labels = self._pad(outs, T5_PAD_ID)
        labels[labels == T5_PAD_ID] = IGNORE_INDEX
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy((ids != T5_PAD_ID).astype(np.int64)),
                "labels": torch.from_numpy(labels), "split": samples.mode}


def temperature_ce(logits: torch.Tensor, labels: torch.Tensor, temperature: float, backward: bool = False):
    """TIGER's loss (model.py:153-161) on the HIP device: mean over labelled positions of CE(logits / temperature, labels),
    labels aligned with the logits (an encoder-decoder: no shift).  Runs gamer_ce_fwd (and gamer_ce_bwd when
    ``backward``): the kernels take the target of row t from labels[t + 1] (the decoder-only shift), so the rows are
    presented as ONE sequence of T + 1 positions whose labels are moved one place to the right; the extra last row has
    no target.  Returns (loss, d loss / d logits or None).  ``logits`` [B, L, V] fp32 on the device is not modified."""
    from . import ops
    if not logits.is_cuda:
        raise RuntimeError("temperature_ce runs on the HIP device only (no CPU fallback)")
    B, L, V = logits.shape
    T = B * L
    ldl = (V + 31) // 32 * 32
    buf = torch.zeros(T + 1, ldl, dtype=torch.float32, device=logits.device)
    buf[:T, :V] = logits.reshape(T, V)
    lab = torch.full((1, T + 1), IGNORE_INDEX, dtype=torch.int64, device=logits.device)
    lab[0, 1:] = labels.reshape(T).to(logits.device)
    f32 = dict(dtype=torch.float32, device=logits.device)
    lse, row_loss = torch.empty(T + 1, **f32), torch.empty(T + 1, **f32)
    loss_sum, count = torch.zeros(1, **f32), torch.zeros(1, **f32)
    ops.ce_fwd(buf, ldl, lab, V, float(temperature), IGNORE_INDEX, lse, row_loss, loss_sum, count)
    loss = loss_sum[0] / count[0]
    grad = None
    if backward:
        ops.ce_bwd(buf, ldl, lab, V, float(temperature), IGNORE_INDEX, lse, count, 0.0, 1.0)
        grad = buf[:T, :V].reshape(B, L, V)
    return loss, grad
