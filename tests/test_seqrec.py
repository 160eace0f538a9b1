"""BASELINE configs[0] (`train_decoder` plumbing: TIGER on a single-behaviour dataset, CPU): SeqRecDataset prefixes,
EncoderDecoderCollator layout and TIGER's temperature loss against the fixture the REAL reference classes produced
(tests/golden/seqrec_small.npz, oracle/make_golden_seqrec.py).  CPU: gamer_amd.seqrec_data and the oracle restatement,
bit-exact on every id tensor.  GPU: the loss / gradient through the HIP loss kernels (gamer_ce_fwd / gamer_ce_bwd)."""
import json
import os

import numpy as np
import pytest
import torch

from gamer_amd import seqrec_data as sq
from oracle import seqrec_oracle as so


@pytest.fixture(scope="module")
def fx(golden, tmp_path_factory):
    z, meta = golden("seqrec_small")
    root = tmp_path_factory.mktemp("seqrec")
    os.makedirs(root / meta["name"])
    for suffix, text in meta["files"].items():
        (root / meta["name"] / (meta["name"] + suffix)).write_text(text)
    return z, meta, str(root)


@pytest.mark.parametrize("case", ["his20", "his3_len10", "his0"])
def test_dataset_and_collator_reproduce_the_reference(fx, case):
    z, meta, root = fx
    c = meta["cases"][case]
    data = sq.SeqRecData(root, meta["name"])
    assert data.new_tokens == [str(t) for t in z[f"{case}/new_tokens"]]
    assert [data.token_id[t] for t in data.new_tokens] == z[f"{case}/new_token_ids"].tolist()
    assert data.vocab_size == int(z[f"{case}/vocab_size"]) and sq.T5_BASE_VOCAB == int(z[f"{case}/base_vocab_size"])
    items, coll_items = data.all_items()
    assert len(items) == int(z[f"{case}/n_all_items"]) and len(coll_items) == int(z[f"{case}/n_collision_items"]) >= 1
    coll = sq.EncoderDecoderCollator(data, model_max_length=c["model_max_length"])
    indices = json.loads(meta["files"][".index.json"])
    inters = json.loads(meta["files"][".inter.json"])
    vocab = so.vocabulary(indices)
    truncated = 0
    for mode in ("train", "valid", "test"):
        s = data.samples(mode, c["max_his_len"])
        ref = so.samples(inters, mode, c["max_his_len"])
        assert len(s) == len(ref) == int(z[f"{case}/{mode}/n"])
        nb = int(z[f"{case}/{mode}/n_batches"])
        for b in range(nb):
            idx = list(range(b * meta["batch"], min(len(s), (b + 1) * meta["batch"])))
            got = coll(s, idx)
            o_ids, o_am, o_lab = so.collate([ref[i] for i in idx], indices, vocab, c["model_max_length"])
            for k, o in (("input_ids", o_ids), ("attention_mask", o_am), ("labels", o_lab)):
                want = z[f"{case}/{mode}/{b}/{k}"]
                assert got[k].dtype == torch.int64 and np.array_equal(got[k].numpy(), want), (case, mode, b, k)
                assert np.array_equal(o, want), ("oracle", case, mode, b, k)
            assert got["split"] == mode
            truncated += int((got["input_ids"].shape[1] == c["model_max_length"]))
    if case == "his3_len10":
        assert truncated > 0            # 3 items x 4 tokens + </s> = 13 > 10: the tokenizer's truncation is exercised


def test_oracle_temperature_ce_matches_tiger(fx):
    z, _, _ = fx
    loss, g = so.temperature_ce(z["tiger/logits"], z["tiger/labels"], float(z["tiger/temperature"]))
    assert abs(loss - float(z["tiger/loss"])) < 2e-6 * float(z["tiger/loss"])
    assert np.abs(g[:, :, ::97] - z["tiger/dlogits_sample"]).max() < 1e-7
    lab = z["tiger/labels"]
    at = np.array([[g[b, t, max(int(lab[b, t]), 0)] for t in range(lab.shape[1])] for b in range(lab.shape[0])])
    assert np.abs(at - z["tiger/dlogits_at_labels"]).max() < 1e-7
    assert (g[lab == -100] == 0).all()


@pytest.mark.gpu
def test_hip_temperature_ce_matches_tiger(fx):
    z, _, _ = fx
    logits = torch.from_numpy(z["tiger/logits"]).cuda()
    labels = torch.from_numpy(z["tiger/labels"]).cuda()
    before = logits.clone()
    loss, grad = sq.temperature_ce(logits, labels, float(z["tiger/temperature"]), backward=True)
    assert torch.equal(logits, before)
    assert abs(float(loss) - float(z["tiger/loss"])) < 2e-6 * float(z["tiger/loss"])
    o_loss, o_g = so.temperature_ce(z["tiger/logits"], z["tiger/labels"], float(z["tiger/temperature"]))
    g = grad.cpu().numpy()
    assert np.abs(g - o_g).max() < 1e-6 * np.abs(o_g).max()
    assert np.abs(g[:, :, ::97] - z["tiger/dlogits_sample"]).max() < 1e-6 * np.abs(o_g).max()
    assert (g[z["tiger/labels"] == -100] == 0).all()
    with pytest.raises(RuntimeError):
        sq.temperature_ce(logits.cpu(), labels, 0.7)
