"""Real-data ingestion for the SMB decoder path: the reference's dataset + tokenizer + collator as a pre-tokeniser
that emits id tensors directly.

The reference builds, per sample, a string such as ``<behavior_click><a_3><b_7><c_1><d_9>...`` and lets a BPE
tokenizer find the added tokens again in every batch (its host bottleneck).  Here every item becomes a row of
token ids once, a sample is an array of item rows, and collation is slicing + padding of flat arrays.  The
results are the reference's tensors, bit for bit (``tests/test_data.py`` against ``tests/golden/data_small.npz``,
generated from the real classes by ``oracle/make_golden_data.py``).

Reference behaviour restated:
  on-disk format            ref:docs/datasets.md:57-94, ref:SeqRec/datasets/SMB_dataset.py:73-147
  session split             SMB_dataset.py:99-115 (ids shifted to start at 0; test = last session, valid = the one
                            before, train = everything before the validation session)
  time                      SMB_dataset.py:116-123 (half hours since the user's first interaction)
  per-token generators      SMB_dataset.py:194-248 (session ids, extended session ids, actions, times)
  training samples          SMB_dataset.py:526-610 (SMBExplicitDatasetForDecoder: one full sequence per user +
                            ``augment`` down-sampled copies drawn from numpy's global stream seeded with 42)
  validation / test samples SMB_dataset.py:272-296, 330-354, 414-443 (filter_by_behavior)
  vocabulary                SMB_dataset.py:356-367 + train_SMB_decoder.py:251 (``tokenizer.add_tokens(sorted(new))``
                            on top of ref:config/s2s-models/Qwen3Multi/vocab.json: ids 14, 15, ... in sorted order)
  collators                 ref:SeqRec/datasets/collator.py:47-107 (train/valid: right padding, pad and behaviour
                            tokens -> -100, validation also masks the history), :146-205 (test: left padding,
                            one extra session id for the behaviour token test_SMB_decoder.py:107-121 appends)
"""
from __future__ import annotations

import datetime
import json
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

# ref:config/s2s-models/Qwen3Multi/vocab.json (the tokenizer every Qwen3* backbone of the reference starts from)
QWEN3MULTI_BASE_VOCAB = {
    "<|vision_end|>": 0, "<|object_ref_start|>": 1, "<|object_ref_end|>": 2, "<|quad_end|>": 3, "<|endoftext|>": 4,
    "<|box_end|>": 5, "<|image_pad|>": 6, "<|box_start|>": 7, "<|im_end|>": 8, "<|im_start|>": 9,
    "<|vision_start|>": 10, "<|video_pad|>": 11, "<|quad_start|>": 12, "<|vision_pad|>": 13,
}
PAD_TOKEN = "<|endoftext|>"
IGNORE_INDEX = -100
ACTION_PAD = 100            # collator.py:99
TIME_PAD = -1.0             # collator.py:105


class TokenTable:
    """token string -> id, exactly what ``Qwen2Tokenizer.from_pretrained(base) + add_tokens(sorted(new_tokens))``
    assigns: the base vocabulary keeps its ids, every new token not already present gets the next free id in
    list order (train_SMB_decoder.py:251)."""

    def __init__(self, new_tokens: Sequence[str], base_vocab: Optional[Dict[str, int]] = None):
        self.vocab: Dict[str, int] = dict(QWEN3MULTI_BASE_VOCAB if base_vocab is None else base_vocab)
        for t in new_tokens:
            if t not in self.vocab:
                self.vocab[t] = len(self.vocab)
        self.pad_id = self.vocab[PAD_TOKEN]

    def __len__(self) -> int:
        return len(self.vocab)

    def __getitem__(self, token: str) -> int:
        return self.vocab[token]


@dataclass
class SampleSet:
    """Samples in flat (CSR) form: sample n owns ``[ptr[n], ptr[n+1])`` of the per-token arrays.

    ``tokens``/``session_ids``/``extended_session_ids``/``actions`` cover history + target for train/valid samples
    and the history only for test samples (the target item is what gets generated); ``n_history`` is the number of
    history TOKENS; ``time`` has its own offsets because the reference's time list is one item shorter."""
    mode: str
    ptr: np.ndarray
    tokens: np.ndarray
    session_ids: np.ndarray
    extended_session_ids: np.ndarray
    actions: np.ndarray
    n_history: np.ndarray
    time_ptr: np.ndarray
    time: np.ndarray
    behavior: list                      # train/valid: behaviour name per sample; test: list of names per sample
    targets: Optional[list] = None      # test: per sample, token-id rows [n_targets, tokens per item] of the target items
    users: Optional[list] = None

    def __len__(self) -> int:
        return len(self.ptr) - 1

    def filter_by_behavior(self, behavior: str) -> "SampleSet":
        """BaseSMBDataset.filter_by_behavior (SMB_dataset.py:414-443)."""
        if self.mode in ("train", "valid"):
            keep = [n for n, b in enumerate(self.behavior) if b == behavior]
            tg = None
            bh = [self.behavior[n] for n in keep]
        else:
            keep = [n for n, bs in enumerate(self.behavior) if behavior in bs]
            tg = [self.targets[n][[i for i, b in enumerate(self.behavior[n]) if b == behavior]] for n in keep]
            bh = [[b for b in self.behavior[n] if b == behavior] for n in keep]
        return self._take(keep, bh, tg)

    def _take(self, keep: List[int], behavior, targets) -> "SampleSet":
        def gather(ptr, *arrays):
            lens = np.array([ptr[n + 1] - ptr[n] for n in keep], dtype=np.int64)
            new_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            idx = (np.concatenate([np.arange(ptr[n], ptr[n + 1]) for n in keep]) if keep else np.zeros(0, np.int64))
            return new_ptr, [a[idx] for a in arrays]
        ptr, (tok, sid, ext, act) = gather(self.ptr, self.tokens, self.session_ids, self.extended_session_ids,
                                           self.actions)
        tptr, (tm,) = gather(self.time_ptr, self.time)
        return SampleSet(self.mode, ptr, tok, sid, ext, act, self.n_history[keep] if keep else self.n_history[:0], tptr,
                         tm, behavior, targets, [self.users[n] for n in keep] if self.users else None)


class SMBData:
    """One dataset directory in the reference's format, split by session (BaseSMBDataset._load_data)."""

    def __init__(self, data_path: str, dataset: str, index_file: str = ".index.json",
                 base_vocab: Optional[Dict[str, int]] = None):
        d = os.path.join(data_path, dataset)

        def load(suffix):
            with open(os.path.join(d, dataset + suffix)) as f:
                return json.load(f)
        self.inters: Dict[str, List[int]] = load(".SMB.inter.json")
        self.behaviors_by_user: Dict[str, List[str]] = load(".SMB.behavior.json")
        raw_session: Dict[str, List[int]] = load(".SMB.session.json")
        raw_time: Dict[str, List[str]] = load(".SMB.time.json")
        self.behavior_level: Dict[str, int] = load(".behavior_level.json")
        indices: Dict[str, List[str]] = load(index_file)
        lens = {len(v) for v in indices.values()}
        if len(lens) != 1:
            raise ValueError(f"All indices must have the same length, but got lengths: {lens}")
        self.sole_item_len = lens.pop()
        self.token_count = self.sole_item_len + 1                     # behaviour token + item tokens
        self.behaviors = list(self.behavior_level.keys())
        self.max_behavior_level = max(self.behavior_level.values())
        top = [b for b, lv in self.behavior_level.items() if lv == self.max_behavior_level]
        if len(top) != 1:
            raise ValueError(f"Expected exactly one target behavior with max level, but found {len(top)}: {top}")
        self.target_behavior = top[0]
        # vocabulary: item tokens + behaviour tokens, sorted (get_new_tokens), appended to the base vocabulary
        new = {t for idx in indices.values() for t in idx}
        new.update(f"<behavior_{b}>" for b in self.behaviors)
        self.new_tokens = sorted(new)
        self.tokens = TokenTable(self.new_tokens, base_vocab)
        self.behavior_token_ids = {b: self.tokens[f"<behavior_{b}>"] for b in self.behaviors}
        # item id -> row of token ids (this replaces the per-batch string tokenisation)
        self.item_keys = list(indices.keys())
        max_item = max(int(k) for k in self.item_keys)
        self.item_tokens = np.full((max_item + 1, self.sole_item_len), -1, dtype=np.int64)
        for k, toks in indices.items():
            self.item_tokens[int(k)] = [self.tokens[t] for t in toks]
        self._indices = indices
        # sessions (ids shifted to start at 0) and the split positions
        self.session: Dict[str, np.ndarray] = {}
        self.test_pos: Dict[str, int] = {}
        self.valid_pos: Dict[str, int] = {}
        for uid, s in raw_session.items():
            s = np.asarray(s, dtype=np.int64)
            s = s - s.min()
            self.session[uid] = s
            uniq = np.unique(s)
            self.test_pos[uid] = int(np.where(s == uniq[-1])[0].min())
            self.valid_pos[uid] = int(np.where(s == uniq[-2])[0].min()) if len(uniq) >= 2 else -1
        fmt = "%Y-%m-%d %H:%M:%S"
        self.time: Dict[str, np.ndarray] = {}
        for uid, ts in raw_time.items():
            t = [datetime.datetime.strptime(x, fmt) for x in ts]
            self.time[uid] = np.array([(x - t[0]).total_seconds() / 1800 for x in t], dtype=np.float64)

    # ---- per-token generators (SMB_dataset.py:175-248) ------------------------------------------------
    def _tokens(self, items: np.ndarray, behaviors: Sequence[str]) -> np.ndarray:
        """[n, token_count] ids of ``<behavior_b><tok1>..<tokK>`` (behavior_first=True, SMB_dataset.py:510-515)"""
        out = np.empty((len(items), self.token_count), dtype=np.int64)
        out[:, 0] = [self.behavior_token_ids[b] for b in behaviors]
        out[:, 1:] = self.item_tokens[items] if len(items) else 0
        return out

    def _session_tokens(self, sids: np.ndarray, keep: int):
        """(_generate_session_ids, _generate_extended_session_ids) of the last ``keep`` items"""
        sids = sids[-keep:] if keep > 0 else sids
        tc = self.token_count
        new = np.ones(len(sids), dtype=bool)
        new[1:] = sids[1:] != sids[:-1]
        rank = np.cumsum(new) - 1
        ext = (rank[:, None] * tc + np.arange(tc)[None, :]).reshape(-1)
        return np.repeat(sids, tc), ext

    def _action_tokens(self, behaviors: Sequence[str], keep: int) -> np.ndarray:
        behaviors = behaviors[-keep:] if keep > 0 else behaviors
        return np.repeat(np.array([self.behavior_level[b] for b in behaviors], dtype=np.int64), self.token_count)

    def _time_tokens(self, times: np.ndarray, max_his_len: int) -> np.ndarray:
        t = np.abs(times - times[-1])
        if max_his_len > 0:
            t = t[-(max_his_len + 1):]
        return np.repeat(t[:-1], self.token_count)

    # ---- sample builders -----------------------------------------------------------------------------
    def _build(self, mode: str, rows: list, max_his_len: int) -> SampleSet:
        """rows: dicts with hist_items, hist_beh, target (item, beh) or None, sids, act_beh, times, behavior, targets"""
        crop_hist = max_his_len if max_his_len > 0 else 0
        crop_full = (max_his_len + 1 if mode in ("train", "valid") else max_his_len) if max_his_len > 0 else 0
        toks, sid, ext, act, tim, nh = [], [], [], [], [], []
        for r in rows:
            hi, hb = r["hist_items"], r["hist_beh"]
            if crop_hist:
                hi, hb = hi[-crop_hist:], hb[-crop_hist:]
            t = self._tokens(np.asarray(hi, dtype=np.int64), hb).reshape(-1)
            nh.append(len(t))
            if r["target"] is not None:
                t = np.concatenate([t, self._tokens(np.asarray([r["target"][0]]), [r["target"][1]]).reshape(-1)])
            s, e = self._session_tokens(np.asarray(r["sids"], dtype=np.int64), crop_full)
            toks.append(t); sid.append(s); ext.append(e)
            act.append(self._action_tokens(r["act_beh"], crop_full))
            tim.append(self._time_tokens(np.asarray(r["times"], dtype=np.float64), max_his_len))
        def cat(parts, dtype):
            return np.concatenate(parts).astype(dtype) if parts else np.zeros(0, dtype)
        lens = np.array([len(t) for t in toks], dtype=np.int64)
        for name, parts in (("session_ids", sid), ("extended_session_ids", ext), ("actions", act)):
            if any(len(p) != n for p, n in zip(parts, lens)):
                raise AssertionError(f"{name}: per-token arrays out of step with the token ids")
        return SampleSet(mode=mode, ptr=np.concatenate([[0], np.cumsum(lens)]).astype(np.int64),
                         tokens=cat(toks, np.int64), session_ids=cat(sid, np.int64),
                         extended_session_ids=cat(ext, np.int64), actions=cat(act, np.int64),
                         n_history=np.asarray(nh, dtype=np.int64),
                         time_ptr=np.concatenate([[0], np.cumsum([len(t) for t in tim])]).astype(np.int64),
                         time=cat(tim, np.float64), behavior=[r["behavior"] for r in rows],
                         targets=[r["targets"] for r in rows] if mode == "test" else None,
                         users=[r["user"] for r in rows])

    def train_samples(self, max_his_len: int, augment: Optional[int] = None) -> SampleSet:
        """SMBExplicitDatasetForDecoder._process_train_data (SMB_dataset.py:586-610): per user with a training
        prefix, the whole prefix as one sequence (last interaction = target) plus ``augment`` copies with the
        lower-level behaviours thinned out (:540-584).  The thinning draws from numpy's legacy global stream seeded
        with 42 (``set_seed(42)``), user by user, so the same RandomState sequence reproduces it."""
        if augment is not None and augment < 1:
            raise ValueError("augment must be greater than or equal to 1")
        rs = np.random.RandomState(42)
        rows = []
        for uid, inter in self.inters.items():
            vp = self.valid_pos[uid]
            if vp <= 0:
                continue
            items = list(inter[:vp])
            behs = list(self.behaviors_by_user[uid][:vp])
            sids = self.session[uid][:vp].tolist()
            times = self.time[uid][:vp].tolist()
            variants = [(items, behs, sids, times)]
            if augment:
                by_beh = {b: [i for i, x in enumerate(behs) if x == b] for b in self.behavior_level}
                for ratio in np.arange(1, augment + 1) / augment:
                    drop: List[int] = []
                    for b, lv in self.behavior_level.items():
                        if lv == self.max_behavior_level or len(by_beh[b]) == 0:
                            continue
                        n_drop = int(len(by_beh[b]) * (ratio / (lv + 1)))
                        if n_drop > 0:
                            drop.extend(rs.choice(by_beh[b], n_drop, replace=False).tolist())
                    mask = np.ones(len(items), dtype=bool)
                    mask[drop] = False
                    if int(mask.sum()) < 2:
                        continue
                    pick = np.nonzero(mask)[0]
                    variants.append(([items[i] for i in pick], [behs[i] for i in pick], [sids[i] for i in pick],
                                     [times[i] for i in pick]))
            for it, bh, sd, tm in variants:
                rows.append(dict(user=uid, hist_items=it[:-1], hist_beh=bh[:-1], target=(it[-1], bh[-1]), sids=sd,
                                 act_beh=bh, times=tm, behavior=bh[-1], targets=None))
        return self._build("train", rows, max_his_len)

    def train_samples_per_interaction(self, max_his_len: int) -> SampleSet:
        """``tasks=smb_explicit`` - BaseSMBDataset._process_train_data (SMB_dataset.py:250-270): every interaction
        of the training prefix except the user's first one is a target; its history is everything BEFORE ITS SESSION
        (empty for the first session), and the session ids / times are those of the session's first interaction."""
        rows = []
        for uid, inter in self.inters.items():
            vp = self.valid_pos[uid]
            if vp <= 0:
                continue
            items, behs = inter[:vp], self.behaviors_by_user[uid][:vp]
            sess, times = self.session[uid], self.time[uid][:vp]
            first = {}
            for idx, sid in enumerate(sess[:vp].tolist()):
                first.setdefault(sid, idx)                         # train_pos (SMB_dataset.py:113-115)
            for i in range(1, len(items)):
                pos = first[int(sess[i])]
                rows.append(dict(user=uid, hist_items=items[:pos], hist_beh=behs[:pos], target=(items[i], behs[i]),
                                 sids=sess[:pos + 1], act_beh=list(behs[:pos]) + [behs[i]], times=times[:pos + 1],
                                 behavior=behs[i], targets=None))
        return self._build("train", rows, max_his_len)

    def valid_samples(self, max_his_len: int) -> SampleSet:
        """BaseSMBDataset._process_valid_data (SMB_dataset.py:272-296): every interaction of the validation session
        is a target after the same history; session ids / times are those of the session's first interaction."""
        rows = []
        for uid, inter in self.inters.items():
            vp = self.valid_pos[uid]
            if vp < 0:
                continue
            tp = self.test_pos[uid]
            items, behs = inter[:tp], self.behaviors_by_user[uid][:tp]
            for i in range(vp, len(items)):
                rows.append(dict(user=uid, hist_items=items[:vp], hist_beh=behs[:vp], target=(items[i], behs[i]),
                                 sids=self.session[uid][:vp + 1], act_beh=list(behs[:vp]) + [behs[i]],
                                 times=self.time[uid][:tp][:vp + 1], behavior=behs[i], targets=None))
        return self._build("valid", rows, max_his_len)

    def test_samples(self, max_his_len: int) -> SampleSet:
        """BaseSMBDataset._process_test_data (SMB_dataset.py:330-354): one sample per user, history = everything
        before the last session, targets = the last session's interactions."""
        rows = []
        for uid, inter in self.inters.items():
            tp = self.test_pos[uid]
            behs = self.behaviors_by_user[uid]
            tb = list(behs[tp:])
            rows.append(dict(user=uid, hist_items=inter[:tp], hist_beh=behs[:tp], target=None,
                             sids=self.session[uid][:tp], act_beh=behs[:tp], times=self.time[uid][:tp + 1], behavior=tb,
                             targets=self.item_tokens[np.asarray(inter[tp:], dtype=np.int64)]))
        return self._build("test", rows, max_his_len)

    def candidate_tokens(self, behavior: str) -> np.ndarray:
        """Token rows of every distinct item under one behaviour, sorted by the item string as
        test_SMB_decoder.py:492-494 sorts ``get_all_items(behavior)``: the input of ``decode.ItemTrie``."""
        beh_token = f"<behavior_{behavior}>"
        by_string = {}
        for k, toks in self._indices.items():
            by_string.setdefault(beh_token + "".join(toks), self.item_tokens[int(k)])
        rows = [np.concatenate([[self.behavior_token_ids[behavior]], by_string[s]]) for s in sorted(by_string)]
        return np.stack(rows).astype(np.int64)


# ---- collation --------------------------------------------------------------------------------------
def _pad_batch(ptr, flat, sel, fill, left: bool, dtype, extra: int = 0):
    """rows ``sel`` of a CSR array as a [len(sel), max_len + extra] matrix padded with ``fill``"""
    lens = (ptr[sel + 1] - ptr[sel]).astype(np.int64)
    L = int(lens.max()) if len(lens) else 0
    out = np.full((len(sel), L + extra), fill, dtype=dtype)
    col = np.arange(L)[None, :]
    if left:
        m = col >= (L - lens)[:, None]
        src = (ptr[sel][:, None] + col - (L - lens)[:, None])[m]
    else:
        m = col < lens[:, None]
        src = (ptr[sel][:, None] + col)[m]
    out[:, :L][m] = flat[src]
    return out, lens, L


class Collator:
    """DecoderOnlyCollator / DecoderOnlyTestCollator on id arrays."""

    def __init__(self, data: SMBData, ignore_behavior_tokens: bool = True, model_max_length: int = 1024):
        self.data = data
        self.pad_id = data.tokens.pad_id
        self.ignore = list(data.behavior_token_ids.values()) if ignore_behavior_tokens else []
        self.model_max_length = model_max_length

    def train(self, samples: SampleSet, index: Sequence[int], only_train_response: bool = False) -> Dict[str, torch.Tensor]:
        """collator.py:55-107: right padding; labels = ids with pad and behaviour tokens -> -100; the history is
        masked as well for a validation batch and with ``only_train_response`` (what every dataset other than
        SMBExplicitDatasetForDecoder gets, train_SMB_decoder.py:271)."""
        sel = np.asarray(index, dtype=np.int64)
        ids, lens, L = _pad_batch(samples.ptr, samples.tokens, sel, self.pad_id, False, np.int64)
        if L > self.model_max_length:
            raise ValueError(f"sequence of {L} tokens exceeds model_max_length={self.model_max_length} "
                             "(the reference truncates silently; lower max_his_len instead)")
        am = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
        labels = ids.copy()
        labels[ids == self.pad_id] = IGNORE_INDEX
        for t in self.ignore:
            labels[ids == t] = IGNORE_INDEX
        if samples.mode == "valid" or only_train_response:
            labels[np.arange(L)[None, :] < samples.n_history[sel][:, None]] = IGNORE_INDEX
        out = {"input_ids": ids, "attention_mask": am, "labels": labels}
        out["session_ids"] = _pad_batch(samples.ptr, samples.session_ids, sel, 0, False, np.int64)[0]
        out["extended_session_ids"] = _pad_batch(samples.ptr, samples.extended_session_ids, sel, 0, False, np.int64)[0]
        out["actions"] = _pad_batch(samples.ptr, samples.actions, sel, ACTION_PAD, False, np.int64)[0]
        out["time"] = _pad_batch(samples.time_ptr, samples.time, sel, TIME_PAD, False, np.float32)[0]
        res = {k: torch.from_numpy(v) for k, v in out.items()}
        res["behavior"] = [samples.behavior[n] for n in sel]
        res["split"] = samples.mode
        return res

    def test(self, samples: SampleSet, index: Sequence[int], behavior: Optional[str] = None):
        """collator.py:146-205 (left padding; session / extended session ids get one more entry, max + 1, for the
        behaviour token that follows).  With ``behavior`` the prompt is completed the way test_SMB_decoder.py:107-121
        does before ``generate``: behaviour token, attention 1 and the behaviour's level appended.
        Returns (inputs, targets) with targets = per sample the token-id rows of the items to be recommended."""
        sel = np.asarray(index, dtype=np.int64)
        extra = 1 if behavior is not None else 0
        ids, lens, L = _pad_batch(samples.ptr, samples.tokens, sel, self.pad_id, True, np.int64, extra)
        am = np.zeros_like(ids)
        am[:, :L] = (np.arange(L)[None, :] >= (L - lens)[:, None])
        act = _pad_batch(samples.ptr, samples.actions, sel, ACTION_PAD, True, np.int64, extra)[0]
        if behavior is not None:
            ids[:, L] = self.data.behavior_token_ids[behavior]
            am[:, L] = 1
            act[:, L] = self.data.behavior_level[behavior]
        sid = _pad_batch(samples.ptr, samples.session_ids, sel, 0, True, np.int64, 1)[0]
        ext = _pad_batch(samples.ptr, samples.extended_session_ids, sel, 0, True, np.int64, 1)[0]
        for row, n in enumerate(sel):
            a, b = samples.ptr[n], samples.ptr[n + 1]
            if b == a:
                raise ValueError("empty history (the reference's test collator fails on it as well, collator.py:181)")
            sid[row, -1] = samples.session_ids[a:b].max() + 1
            ext[row, -1] = samples.extended_session_ids[a:b].max() + 1
        inputs = {"input_ids": ids, "attention_mask": am, "session_ids": sid, "extended_session_ids": ext, "actions": act}
        inputs = {k: torch.from_numpy(v) for k, v in inputs.items()}
        inputs["behavior"] = [samples.behavior[n] for n in sel]
        return inputs, [samples.targets[n] for n in sel]


def model_config(data: SMBData, max_his_len: int, **overrides):
    """The config mutation of train_SMB_decoder.py:251-252, 321-360 for Qwen3Multi / Qwen3SessionMulti: vocabulary
    size of the extended tokenizer, behaviour token id -> behaviour index in ``behaviors`` order, tokens per item,
    one expert per token position (+1 for pad / eos), n_positions = max_his_len + 1."""
    from .config import Qwen3MultiConfig
    kw = dict(vocab_size=len(data.tokens), num_behavior=len(data.behaviors),
              behavior_maps={str(data.behavior_token_ids[b]): i for i, b in enumerate(data.behaviors)},
              num_positions=data.token_count, num_experts=data.token_count + 1, n_positions=max_his_len + 1,
              pad_token_id=data.tokens.pad_id)
    kw.update(overrides)
    return Qwen3MultiConfig(**kw)


def batches(n: int, batch_size: int, shuffle_seed: Optional[int] = None, rank: int = 0, world: int = 1):
    """index lists for one epoch; with ``world`` > 1 each rank takes every world-th batch (data-parallel shards)"""
    order = np.arange(n)
    if shuffle_seed is not None:
        order = np.random.RandomState(shuffle_seed).permutation(n)
    chunks = [order[i:i + batch_size] for i in range(0, n, batch_size)]
    return chunks[rank::world]
