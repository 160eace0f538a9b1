#!/usr/bin/env python3
"""Golden fixture for the data path, generated from the REAL reference
(build container only): a small synthetic dataset in the reference's on-disk format is pushed through
``SMBExplicitDatasetForDecoder`` (train, augment 2) / ``SMBExplicitDataset`` (valid, test), the reference's
``Qwen2Tokenizer`` (files under ref:config/s2s-models/Qwen3Multi) and its ``DecoderOnlyCollator`` /
``DecoderOnlyTestCollator``.  The fixture holds the dataset files (data, written by
``gamer_amd.synthetic.write_smb_dataset``) and every tensor the reference produced.

Usage:  python oracle/make_golden_data.py        (writes tests/golden/data_small.npz)
"""
import json
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import _ref_loader  # noqa: E402
from gamer_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "data_small.npz")
NAME = "Syn"
MAX_HIS_LEN = 6          # short enough that most users get cropped
AUGMENT = 2
BATCH = 7


def main():
    _ref_loader.load_reference_classes()          # installs the shims + package stubs
    from SeqRec.datasets.SMB_dataset import SMBExplicitDataset, SMBExplicitDatasetForDecoder
    from SeqRec.datasets.collator import DecoderOnlyCollator, DecoderOnlyTestCollator
    from transformers import Qwen2Tokenizer

    tmp = os.path.join(ROOT, "oracle", "_ref", "tmpdata")
    shutil.rmtree(tmp, ignore_errors=True)
    d = synthetic.write_smb_dataset(tmp, NAME, n_users=40, n_items=60, codebook=16, seed=3)
    files = {fn[len(NAME):]: open(os.path.join(d, fn)).read() for fn in sorted(os.listdir(d)) if fn.endswith(".json")}
    kw = dict(dataset=NAME, data_path=tmp, max_his_len=MAX_HIS_LEN, index_file=".index.json", behavior_first=True)
    train = SMBExplicitDatasetForDecoder(mode="train", augment=AUGMENT, **kw)
    valid = SMBExplicitDataset(mode="valid", **kw)
    test = SMBExplicitDataset(mode="test", **kw)

    # Shim for the installed transformers 5.x (the reference pins 4.51, requirements.txt:9): its tokenizers-backed
    # Qwen2Tokenizer refuses the reference's full merges.txt next to the 14-entry vocab.json ("Token out of
    # vocabulary"), the 4.51 slow tokenizer does not check.  Every string on this path consists of ADDED tokens only
    # (<a_i>, <behavior_x>, ...), which are split off before BPE, so the merges are never applied: load the
    # reference's vocab.json + tokenizer_config.json with an empty merge list.
    tdir = os.path.join(tmp, "_tokenizer")
    os.makedirs(tdir, exist_ok=True)
    src = os.path.join(_ref_loader.REF_ROOT, "config", "s2s-models", "Qwen3Multi")
    for fn in ("vocab.json", "tokenizer_config.json"):
        shutil.copy(os.path.join(src, fn), os.path.join(tdir, fn))
    with open(os.path.join(tdir, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    tok = Qwen2Tokenizer.from_pretrained(tdir, model_max_length=1024)
    new_tokens = train.get_new_tokens()
    tok.add_tokens(new_tokens)                                   # train_SMB_decoder.py:251
    behavior_tokens = []
    for b in train.behaviors:
        behavior_tokens.extend(train.get_behavior_tokens(b))
    behavior_ids = [tok.encode(b, add_special_tokens=False)[0] for b in behavior_tokens]      # :263-267
    res = dict(
        new_tokens=np.array(new_tokens), new_token_ids=np.array(tok.convert_tokens_to_ids(new_tokens)),
        vocab_size=np.int64(len(tok)), pad_id=np.int64(tok.pad_token_id), behaviors=np.array(train.behaviors),
        behavior_token_ids=np.array(behavior_ids), target_behavior=np.array(train.target_behavior),
        n_train=np.int64(len(train)), n_valid=np.int64(len(valid)), n_test=np.int64(len(test)),
    )
    # ---- training / validation batches through DecoderOnlyCollator (train_SMB_decoder.py:269-271) ----
    coll = DecoderOnlyCollator(tok, only_train_response=False, ignore_behavior_tokens=behavior_ids)
    for name, ds in (("train", train), ("valid", valid)):
        nb = 0
        for b0 in range(0, len(ds), BATCH):
            batch = coll([ds[i] for i in range(b0, min(len(ds), b0 + BATCH))])
            for k in ("input_ids", "attention_mask", "labels", "session_ids", "extended_session_ids", "actions", "time"):
                res[f"{name}/{nb}/{k}"] = batch[k].numpy()
            res[f"{name}/{nb}/behavior"] = np.array(batch["behavior"])
            assert batch["split"] == name
            nb += 1
        res[f"n_{name}_batches"] = np.int64(nb)
    # ---- tasks=smb_explicit: one sample per interaction, only the response is trained (only_train_response=True) ----
    per_inter = SMBExplicitDataset(mode="train", **kw)
    coll_r = DecoderOnlyCollator(tok, only_train_response=True, ignore_behavior_tokens=behavior_ids)
    res["n_train_per_interaction"] = np.int64(len(per_inter))
    nb = 0
    for b0 in range(0, len(per_inter), 4 * BATCH):
        batch = coll_r([per_inter[i] for i in range(b0, min(len(per_inter), b0 + 4 * BATCH))])
        for k in ("input_ids", "attention_mask", "labels", "session_ids", "extended_session_ids", "actions", "time"):
            res[f"train_pi/{nb}/{k}"] = batch[k].numpy()
        nb += 1
    res["n_train_pi_batches"] = np.int64(nb)
    # ---- test batches through DecoderOnlyTestCollator (test_SMB_decoder.py:430-440), per behaviour ----
    tcoll = DecoderOnlyTestCollator(tok)
    for beh in test.behaviors:
        sub = test.filter_by_behavior(beh)
        res[f"test/{beh}/n"] = np.int64(len(sub))
        nb = 0
        for b0 in range(0, len(sub), BATCH):
            inputs, targets = tcoll([sub[i] for i in range(b0, min(len(sub), b0 + BATCH))])
            for k in ("input_ids", "attention_mask", "session_ids", "extended_session_ids", "actions"):
                res[f"test/{beh}/{nb}/{k}"] = inputs[k].numpy()
            res[f"test/{beh}/{nb}/targets_json"] = np.array(json.dumps(targets))
            res[f"test/{beh}/{nb}/target_ids_json"] = np.array(json.dumps(
                [[tok.encode(t, add_special_tokens=False) for t in ts] for ts in targets]))
            nb += 1
        res[f"test/{beh}/n_batches"] = np.int64(nb)
        # candidate token lists of the constrained decoding (test_SMB_decoder.py:492-494)
        items = sorted(test.get_all_items(beh))
        res[f"test/{beh}/candidates"] = np.array([tok.encode(it, add_special_tokens=False) for it in items])
    meta = dict(name=NAME, max_his_len=MAX_HIS_LEN, augment=AUGMENT, batch=BATCH, files=files,
                generator=dict(reference="wzf2000/GAMER @ /root/reference", transformers=__import__("transformers").__version__,
                               numpy=np.__version__))
    res["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(OUT, **res)
    shutil.rmtree(tmp, ignore_errors=True)
    print(f"train {len(train)} valid {len(valid)} test {len(test)} vocab {len(tok)} -> {OUT} "
          f"({os.path.getsize(OUT) / 1e3:.1f} kB)")
    print("behavior ids", dict(zip(behavior_tokens, behavior_ids)), "first new tokens", new_tokens[:6])


if __name__ == "__main__":
    torch.set_num_threads(4)
    main()
