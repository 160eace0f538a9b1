"""Synthetic SMB-decoder batches with the layout the reference's data path produces.

The reference's datasets are git-LFS pointers, so every parity and
throughput run uses synthetic sequences with the *format* of
``SMBExplicitDatasetForDecoder`` + ``DecoderOnlyCollator``
(ref:SeqRec/datasets/SMB_dataset.py:194-248,586-610; ref:SeqRec/datasets/collator.py:55-107):

* item = ``<behavior_b><a_i><b_j><c_k><d_l>`` -> 5 tokens; sequence = n items, right padded
  with the pad id (4); no BOS/EOS appended;
* token ids: 0..13 specials, semantic code (level l, code c) -> 14 + cb*l + c,
  behaviour b -> 14 + 4*cb + b (cb = codebook size, 256 in the shipped recipe);
* ``actions`` = behaviour *level* of the token's item (here level == b), padded with 100;
* ``labels`` = input_ids with pad and behaviour tokens replaced by -100
  (collator.py:68-73 with ignore_behavior_tokens);
* ``session_ids`` / ``extended_session_ids``: unused by Qwen3Multi (one session per item by default); with
  ``session_mean`` the items are grouped into sessions of random length the way
  ``_generate_session_ids`` / ``_generate_extended_session_ids`` (SMB_dataset.py:194-222) lay them out:
  the raw session id repeated over the item's 5 tokens, and 5 * (session rank) + (token in item) as the
  RoPE position, both right padded with 0.  This is what Qwen3SessionMulti consumes.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

PAD_ID = 4
EOS_ID = 8
N_SPECIAL = 14
TOKENS_PER_ITEM = 5


def vocab_size(codebook: int = 256, num_behavior: int = 3) -> int:
    return N_SPECIAL + 4 * codebook + num_behavior


def behavior_token(b: int, codebook: int = 256) -> int:
    return N_SPECIAL + 4 * codebook + b


def behavior_maps(codebook: int = 256, num_behavior: int = 3) -> Dict[int, int]:
    """token id -> behaviour index, what train_SMB_decoder.py:321-333 stores in the config."""
    return {behavior_token(b, codebook): b for b in range(num_behavior)}


def make_batch(batch_size: int, n_items: int = 101, codebook: int = 256, num_behavior: int = 3,
               ragged: bool = False, min_items: int = 2, seed: int = 20251114,
               behavior_probs: Optional[Sequence[float]] = None,
               pad_rows: Optional[Dict[int, int]] = None,
               session_mean: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """One collated batch (CPU int64 tensors).

    ``ragged``: n_items ~ U{min_items..n_items} per row, right padded to the batch max.
    ``pad_rows``: {row: items_to_drop} for hand-placed padding (used by the golden fixtures).
    ``behavior_probs``: categorical over behaviours (default uniform).
    ``session_mean``: mean number of items per session (None: every item is its own session).
    """
    g = torch.Generator().manual_seed(seed)
    if behavior_probs is None:
        behavior_probs = [1.0 / num_behavior] * num_behavior
    probs = torch.tensor(behavior_probs, dtype=torch.float64)
    lens = torch.full((batch_size,), n_items, dtype=torch.int64)
    if ragged:
        lens = torch.randint(min_items, n_items + 1, (batch_size,), generator=g)
    if pad_rows:
        for r, drop in pad_rows.items():
            lens[r] = max(1, n_items - drop)
    max_items = int(lens.max())
    S = max_items * TOKENS_PER_ITEM
    beh = torch.multinomial(probs.expand(batch_size, -1), max_items, replacement=True, generator=g)
    codes = torch.randint(0, codebook, (batch_size, max_items, 4), generator=g)
    item_tok = torch.empty(batch_size, max_items, TOKENS_PER_ITEM, dtype=torch.int64)
    item_tok[:, :, 0] = N_SPECIAL + 4 * codebook + beh
    for l in range(4):
        item_tok[:, :, 1 + l] = N_SPECIAL + codebook * l + codes[:, :, l]
    ids = item_tok.reshape(batch_size, S)
    actions = beh.repeat_interleave(TOKENS_PER_ITEM, dim=1)
    valid = (torch.arange(max_items)[None, :] < lens[:, None]).repeat_interleave(TOKENS_PER_ITEM, dim=1)
    ids = torch.where(valid, ids, torch.full_like(ids, PAD_ID))
    actions = torch.where(valid, actions, torch.full_like(actions, 100))
    labels = ids.clone()
    labels[~valid] = -100
    labels[:, ::TOKENS_PER_ITEM] = -100          # behaviour tokens are not predicted
    item_index = torch.arange(S) // TOKENS_PER_ITEM
    if session_mean is None:
        session = torch.where(valid, item_index[None, :].expand(batch_size, S), torch.zeros_like(ids))
        ext = torch.where(valid, torch.arange(S)[None, :].expand(batch_size, S), torch.zeros_like(ids))
    else:
        gs = torch.Generator().manual_seed(seed + 7919)      # own stream: the token draws above stay as they were
        new = torch.rand(batch_size, max_items, generator=gs) < (1.0 / float(session_mean))
        new[:, 0] = True
        rank = torch.cumsum(new.to(torch.int64), dim=1) - 1                          # session rank of every item
        base = torch.randint(0, 4, (batch_size, 1), generator=gs)                    # raw ids need not start at 0
        gap = torch.randint(1, 3, (batch_size, max_items), generator=gs) * new       # nor be consecutive
        raw = base + torch.cumsum(gap, dim=1) - gap[:, :1]
        session = torch.where(valid, raw.repeat_interleave(TOKENS_PER_ITEM, dim=1), torch.zeros_like(ids))
        tok_in_item = (torch.arange(S) % TOKENS_PER_ITEM)[None, :]
        ext = torch.where(valid, rank.repeat_interleave(TOKENS_PER_ITEM, dim=1) * TOKENS_PER_ITEM + tok_in_item,
                          torch.zeros_like(ids))
    return {
        "input_ids": ids.contiguous(),
        "attention_mask": valid.to(torch.int64).contiguous(),
        "actions": actions.contiguous(),
        "labels": labels.contiguous(),
        "session_ids": session.contiguous(),
        "extended_session_ids": ext.contiguous(),
    }


def make_augmented_batch(batch_size: int, n_items: int = 101, codebook: int = 256, num_behavior: int = 3, augment: int = 4,
                         seed: int = 20251114, behavior_probs: Optional[Sequence[float]] = None, min_full: int = 20,
                         max_full: int = 250) -> Dict[str, torch.Tensor]:
    """One collated training batch shaped like ``tasks=smb_explicit_decoder_<augment>`` (BASELINE configs[3]): every row is a
    user's interaction list or one of its ``augment`` thinned copies - copy r drops floor(count_b * (r / augment) / (level_b + 1))
    random interactions of every behaviour below the target level (ref:SeqRec/datasets/SMB_dataset.py:540-584) -, cropped
    to the LAST ``n_items`` interactions (max_his_len + 1) and right padded to the batch's longest row, as the collator does.
    Full lengths are U{min_full..max_full} (the real ShortVideoAD statistics are LFS pointers here: a documented guess);
    behaviours i.i.d. ``behavior_probs`` with level == behaviour index."""
    g = torch.Generator().manual_seed(seed)
    if behavior_probs is None:
        behavior_probs = [1.0 / num_behavior] * num_behavior
    probs = torch.tensor(behavior_probs, dtype=torch.float64)
    rows = []
    for _ in range(batch_size):
        n_full = int(torch.randint(min_full, max_full + 1, (1,), generator=g))
        beh = torch.multinomial(probs, n_full, replacement=True, generator=g)
        codes = torch.randint(0, codebook, (n_full, 4), generator=g)
        r = int(torch.randint(0, augment + 1, (1,), generator=g))          # 0 = the unthinned sequence
        keep = torch.ones(n_full, dtype=torch.bool)
        if r:
            for b in range(num_behavior - 1):                              # the highest level is the target behaviour: kept
                idx = torch.nonzero(beh == b).flatten()
                n_drop = int(len(idx) * ((r / augment) / (b + 1)))
                if n_drop > 0:
                    keep[idx[torch.randperm(len(idx), generator=g)[:n_drop]]] = False
            if int(keep.sum()) < 2:
                keep[:] = True
        beh, codes = beh[keep][-n_items:], codes[keep][-n_items:]
        rows.append((beh, codes))
    max_items = max(len(b) for b, _ in rows)
    S = max_items * TOKENS_PER_ITEM
    ids = torch.full((batch_size, S), PAD_ID, dtype=torch.int64)
    actions = torch.full((batch_size, S), 100, dtype=torch.int64)
    valid = torch.zeros(batch_size, S, dtype=torch.bool)
    for i, (beh, codes) in enumerate(rows):
        n = len(beh)
        tok = torch.empty(n, TOKENS_PER_ITEM, dtype=torch.int64)
        tok[:, 0] = N_SPECIAL + 4 * codebook + beh
        for l in range(4):
            tok[:, 1 + l] = N_SPECIAL + codebook * l + codes[:, l]
        ids[i, :n * TOKENS_PER_ITEM] = tok.reshape(-1)
        actions[i, :n * TOKENS_PER_ITEM] = beh.repeat_interleave(TOKENS_PER_ITEM)
        valid[i, :n * TOKENS_PER_ITEM] = True
    labels = ids.clone()
    labels[~valid] = -100
    labels[:, ::TOKENS_PER_ITEM] = -100
    pos = torch.arange(S)[None, :].expand(batch_size, S)
    return {"input_ids": ids, "attention_mask": valid.to(torch.int64), "actions": actions, "labels": labels,
            "session_ids": torch.where(valid, pos // TOKENS_PER_ITEM, torch.zeros_like(ids)),
            "extended_session_ids": torch.where(valid, pos, torch.zeros_like(ids))}


def make_catalogue(n_items: int, codebook: int = 256, seed: int = 7) -> torch.Tensor:
    """``n_items`` distinct semantic-ID tuples [n_items, 4] (the item universe of an evaluation run)."""
    g = torch.Generator().manual_seed(seed)
    seen, rows = set(), []
    while len(rows) < n_items:
        c = tuple(torch.randint(0, codebook, (4,), generator=g).tolist())
        if c not in seen:
            seen.add(c)
            rows.append(c)
    return torch.tensor(rows, dtype=torch.int64)


def item_tokens(codes: torch.Tensor, behavior: int, codebook: int = 256) -> torch.Tensor:
    """[n,4] semantic codes -> [n,5] token ids ``<behavior_b><a><b><c><d>``."""
    n = codes.shape[0]
    tok = torch.empty(n, TOKENS_PER_ITEM, dtype=torch.int64)
    tok[:, 0] = behavior_token(behavior, codebook)
    for l in range(4):
        tok[:, 1 + l] = N_SPECIAL + codebook * l + codes[:, l]
    return tok


def make_eval_batch(batch_size: int, max_his: int, catalogue: torch.Tensor, target_behavior: int, codebook: int = 256,
                    num_behavior: int = 3, min_his: int = 1, seed: int = 1,
                    behavior_probs: Optional[Sequence[float]] = None,
                    session_mean: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Evaluation prompts with the layout of ``DecoderOnlyTestCollator`` + ``test_single_behavior``
    (ref:SeqRec/datasets/collator.py:149-207, ref:SeqRec/tasks/test_SMB_decoder.py:98-118): histories of
    catalogue items, LEFT padded (ids with the pad id, actions with 100), followed by the target behaviour
    token.  ``targets`` [B,4] are the semantic codes of the held-out item."""
    g = torch.Generator().manual_seed(seed)
    if behavior_probs is None:
        behavior_probs = [1.0 / num_behavior] * num_behavior
    probs = torch.tensor(behavior_probs, dtype=torch.float64)
    lens = torch.randint(min_his, max_his + 1, (batch_size,), generator=g)
    lens[0] = max_his
    S = max_his * TOKENS_PER_ITEM + 1
    ids = torch.full((batch_size, S), PAD_ID, dtype=torch.int64)
    actions = torch.full((batch_size, S), 100, dtype=torch.int64)
    mask = torch.zeros(batch_size, S, dtype=torch.int64)
    targets = torch.empty(batch_size, 4, dtype=torch.int64)
    for b in range(batch_size):
        n = int(lens[b])
        pick = torch.randint(0, catalogue.shape[0], (n + 1,), generator=g)
        beh = torch.multinomial(probs, n, replacement=True, generator=g)
        start = S - 1 - n * TOKENS_PER_ITEM
        for i in range(n):
            ids[b, start + 5 * i: start + 5 * i + 5] = item_tokens(catalogue[pick[i]][None], int(beh[i]), codebook)[0]
            actions[b, start + 5 * i: start + 5 * i + 5] = int(beh[i])
        ids[b, S - 1] = behavior_token(target_behavior, codebook)
        actions[b, S - 1] = target_behavior
        mask[b, start:] = 1
        targets[b] = catalogue[pick[n]]
    pos = torch.arange(S)[None, :].expand(batch_size, S)
    if session_mean is None:
        return {"input_ids": ids, "attention_mask": mask, "actions": actions, "targets": targets,
                "session_ids": torch.where(mask.bool(), pos // TOKENS_PER_ITEM, torch.zeros_like(ids)),
                "extended_session_ids": torch.where(mask.bool(), pos, torch.zeros_like(ids))}
    # Layout of DecoderOnlyTestCollator for the session models (collator.py:176-195): the history's session ids
    # (0-based, SMB_dataset.py:106) and extended ids, LEFT padded with 0, then max + 1 for the behaviour token
    gs = torch.Generator().manual_seed(seed + 7919)
    sess = torch.zeros(batch_size, S, dtype=torch.int64)
    ext = torch.zeros(batch_size, S, dtype=torch.int64)
    for b in range(batch_size):
        n = int(lens[b])
        new = torch.rand(n, generator=gs) < (1.0 / float(session_mean))
        new[0] = True
        gap = torch.randint(1, 3, (n,), generator=gs) * new
        raw = torch.cumsum(gap, 0) - gap[0]
        rank = torch.cumsum(new.to(torch.int64), 0) - 1
        start = S - 1 - n * TOKENS_PER_ITEM
        sess[b, start:S - 1] = raw.repeat_interleave(TOKENS_PER_ITEM)
        ext[b, start:S - 1] = (rank.repeat_interleave(TOKENS_PER_ITEM) * TOKENS_PER_ITEM
                               + torch.arange(TOKENS_PER_ITEM).repeat(n))
        sess[b, S - 1] = int(raw.max()) + 1
        ext[b, S - 1] = int(ext[b].max()) + 1
    return {"input_ids": ids, "attention_mask": mask, "actions": actions, "targets": targets,
            "session_ids": sess, "extended_session_ids": ext}


def write_smb_dataset(root: str, name: str, n_users: int = 40, n_items: int = 60, codebook: int = 16,
                      behavior_level: Optional[Dict[str, int]] = None, max_sessions: int = 7,
                      max_per_session: int = 4, seed: int = 0, min_sessions: int = 2) -> str:
    """Write a small session-wise multi-behaviour dataset in the reference's ON-DISK format
    (ref:docs/datasets.md:57-94; the shipped data files are git-LFS pointers): ``<root>/<name>/<name>.SMB.inter.json``
    (user -> item ids), ``.SMB.behavior.json`` (user -> behaviour names), ``.SMB.session.json`` (user -> raw session
    ids, not 0-based), ``.SMB.time.json`` (user -> "%Y-%m-%d %H:%M:%S"), ``.behavior_level.json``, ``.index.json``
    (item -> 4 semantic-ID tokens).  Users get min_sessions..max_sessions sessions: with two sessions a user has no
    training prefix (skipped by SMB_dataset.py:567-568) and an empty validation history; single-session users
    (min_sessions=1) make the reference's test collator fail on an empty history (collator.py:181), so the default
    leaves them out.  Returns the dataset directory."""
    import datetime
    import json
    import os
    import random
    rng = random.Random(seed)
    if behavior_level is None:
        behavior_level = {"click": 0, "cart": 1, "buy": 2}
    names = list(behavior_level)
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    index = {str(i): [f"<{c}_{rng.randrange(codebook)}>" for c in "abcd"] for i in range(n_items)}
    inter, beh, sess, tim = {}, {}, {}, {}
    for u in range(n_users):
        n_sess = rng.randint(min_sessions, max_sessions)
        sid = rng.randint(0, 5)
        t = datetime.datetime(2024, 1, 1) + datetime.timedelta(minutes=rng.randint(0, 10 ** 5))
        items, bs, ss, ts = [], [], [], []
        for _ in range(n_sess):
            for _ in range(rng.randint(1, max_per_session)):
                items.append(rng.randrange(n_items))
                bs.append(rng.choices(names, weights=[6, 2, 1][:len(names)] + [1] * max(0, len(names) - 3))[0])
                ss.append(sid)
                t += datetime.timedelta(seconds=rng.randint(30, 4000))
                ts.append(t.strftime("%Y-%m-%d %H:%M:%S"))
            sid += rng.randint(1, 3)
            t += datetime.timedelta(hours=rng.randint(2, 60))
        inter[str(u)], beh[str(u)], sess[str(u)], tim[str(u)] = items, bs, ss, ts
    for suffix, obj in ((".SMB.inter.json", inter), (".SMB.behavior.json", beh), (".SMB.session.json", sess),
                        (".SMB.time.json", tim), (".behavior_level.json", behavior_level), (".index.json", index)):
        with open(os.path.join(d, name + suffix), "w") as f:
            json.dump(obj, f)
    return d
