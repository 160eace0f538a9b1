#!/usr/bin/env python3
"""Golden fixture for the `train_decoder` plumbing (BASELINE configs[0]), generated from the
REAL reference (build container only): a small dataset in the reference's on-disk format goes through ``SeqRecDataset``
(train / valid / test), the reference's ``T5Tokenizer`` (ref:config/s2s-models/TIGER) with the dataset's tokens added,
``EncoderDecoderCollator`` and one forward of ``TIGER`` (temperature 0.7) whose logits / labels / loss pin the
temperature cross-entropy.  The fixture holds the dataset files (data) and every tensor the reference produced.

Two shims for the installed transformers 5.x (the reference pins 4.51, requirements.txt:9), applied here and nowhere
else: ``modeling_t5.__HEAD_MASK_WARNING_MSG`` (a message string TIGER/model.py:5 imports, removed in 5.x) and the
``model_parallel`` attribute (T5's deprecated naive pipeline flag, TIGER/model.py:105; False).

Usage:  python oracle/make_golden_seqrec.py        (writes tests/golden/seqrec_small.npz)
"""
import importlib.machinery
import json
import os
import random
import shutil
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import _ref_loader  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "seqrec_small.npz")
NAME = "Syn"
BATCH = 6


def write_dataset(root: str, seed: int = 5):
    rnd = random.Random(seed)
    os.makedirs(os.path.join(root, NAME), exist_ok=True)
    # 4 semantic-id tokens per item (TIGER's RQ codes), a few items share a prefix, one pair collides completely
    index = {str(i): [f"<a_{rnd.randrange(12)}>", f"<b_{rnd.randrange(12)}>", f"<c_{rnd.randrange(12)}>", f"<d_{rnd.randrange(4)}>"]
             for i in range(40)}
    index["39"] = list(index["7"])
    inter = {str(u): [rnd.randrange(40) for _ in range(rnd.randrange(3, 14))] for u in range(24)}
    files = {".index.json": json.dumps(index), ".inter.json": json.dumps(inter)}
    for suffix, text in files.items():
        with open(os.path.join(root, NAME, NAME + suffix), "w") as f:
            f.write(text)
    return files


def main():
    _ref_loader._install_shims()
    for parent in ("SeqRec", "SeqRec.models", "SeqRec.models.generative", "SeqRec.datasets"):
        if parent not in sys.modules:
            pkg = types.ModuleType(parent)
            pkg.__path__ = [os.path.join(_ref_loader.REF_ROOT, *parent.split("."))]
            pkg.__spec__ = importlib.machinery.ModuleSpec(parent, None, is_package=True)
            pkg.__spec__.submodule_search_locations = pkg.__path__
            sys.modules[parent] = pkg
    import transformers.models.t5.modeling_t5 as mt5
    if not hasattr(mt5, "__HEAD_MASK_WARNING_MSG"):
        setattr(mt5, "__HEAD_MASK_WARNING_MSG", "")
    from transformers import T5Config, T5Tokenizer
    from SeqRec.datasets.collator import EncoderDecoderCollator
    from SeqRec.datasets.seq_dataset import SeqRecDataset
    from SeqRec.models.generative.TIGER.model import TIGER

    tmp = os.path.join(ROOT, "oracle", "_ref", "tmpseq")
    shutil.rmtree(tmp, ignore_errors=True)
    files = write_dataset(tmp)
    base = os.path.join(_ref_loader.REF_ROOT, "config", "s2s-models", "TIGER")
    res = {}
    cases = {"his20": dict(max_his_len=20, model_max_length=512),          # BASELINE configs[0]: max_his_len=20
             "his3_len10": dict(max_his_len=3, model_max_length=10),       # cropped histories AND tokenizer truncation (13 -> 9 + </s>)
             "his0": dict(max_his_len=0, model_max_length=512)}            # max_his_len <= 0: no cropping
    tok = None
    for cname, c in cases.items():
        tok = T5Tokenizer.from_pretrained(base, model_max_length=c["model_max_length"], legacy=True)   # train_decoder.py:193-197
        n_base = len(tok)
        for mode in ("train", "valid", "test"):
            ds = SeqRecDataset(dataset=NAME, data_path=tmp, max_his_len=c["max_his_len"], index_file=".index.json", mode=mode)
            if mode == "train":
                new_tokens = ds.get_new_tokens()
                tok.add_tokens(new_tokens)                                                      # train_decoder.py:239
                res[f"{cname}/new_tokens"] = np.array(new_tokens)
                res[f"{cname}/new_token_ids"] = np.array(tok.convert_tokens_to_ids(new_tokens))
                res[f"{cname}/vocab_size"] = np.int64(len(tok))
                res[f"{cname}/base_vocab_size"] = np.int64(n_base)
                items = ds.get_all_items()
                res[f"{cname}/n_all_items"] = np.int64(len(items))
                res[f"{cname}/n_collision_items"] = np.int64(len(ds.collision_items))
            coll = EncoderDecoderCollator(tok)
            res[f"{cname}/{mode}/n"] = np.int64(len(ds))
            nb = 0
            for b0 in range(0, len(ds), BATCH):
                batch = coll([ds[i] for i in range(b0, min(len(ds), b0 + BATCH))])
                for k in ("input_ids", "attention_mask", "labels"):
                    res[f"{cname}/{mode}/{nb}/{k}"] = batch[k].numpy()
                assert batch["split"] == mode
                nb += 1
            res[f"{cname}/{mode}/n_batches"] = np.int64(nb)
    # ---- TIGER forward on the first training batch of the configs[0] case: logits, labels, temperature loss ----
    cfg = T5Config.from_pretrained(base)
    cfg.vocab_size = int(res["his20/vocab_size"])                                               # train_decoder.py:240
    torch.manual_seed(0)
    model = TIGER(cfg)
    model.model_parallel = False
    model.set_hyper(0.7)
    model.resize_token_embeddings(int(res["his20/vocab_size"]))
    model.eval()
    ids = torch.from_numpy(res["his20/train/0/input_ids"])[:3]
    am = torch.from_numpy(res["his20/train/0/attention_mask"])[:3]
    lab = torch.from_numpy(res["his20/train/0/labels"])[:3].clone()
    lab[2, 3:] = -100            # a shorter target, as the collator pads them (items with fewer tokens)
    out = model(input_ids=ids, attention_mask=am, labels=lab)
    logits = out.logits.detach().clone().requires_grad_(True)
    # d loss / d logits of the reference's own loss expression (model.py:155-159) on those logits
    loss2 = torch.nn.CrossEntropyLoss(ignore_index=-100)((logits / 0.7).view(-1, logits.size(-1)), lab.view(-1))
    loss2.backward()
    assert float(loss2) == float(out.loss)
    res["tiger/logits"] = out.logits.detach().numpy()
    res["tiger/labels"] = lab.numpy()
    res["tiger/loss"] = np.float64(float(out.loss))
    res["tiger/temperature"] = np.float64(0.7)
    g = logits.grad.numpy()
    res["tiger/dlogits_sample"] = g[:, :, ::97].copy()
    res["tiger/dlogits_at_labels"] = np.array([[g[b, t, max(int(lab[b, t]), 0)] for t in range(lab.shape[1])] for b in range(lab.shape[0])])
    meta = dict(name=NAME, batch=BATCH, files=files, cases=cases,
                generator=dict(reference="wzf2000/GAMER @ /root/reference", transformers=__import__("transformers").__version__,
                               torch=torch.__version__))
    res["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(OUT, **res)
    shutil.rmtree(tmp, ignore_errors=True)
    print({k: int(res[f"{k}/train/n"]) for k in cases}, "vocab", int(res["his20/vocab_size"]), "loss", float(out.loss),
          f"-> {OUT} ({os.path.getsize(OUT) / 1e3:.1f} kB)")


if __name__ == "__main__":
    torch.set_num_threads(4)
    main()
