"""Data path: gamer_amd.data against tensors produced by the reference's own
SMBExplicitDatasetForDecoder / SMBExplicitDataset + Qwen2Tokenizer + DecoderOnlyCollator / DecoderOnlyTestCollator
(tests/golden/data_small.npz, oracle/make_golden_data.py).  Integer work: bit-exact; times: equal float32."""
import json
import os

import numpy as np
import pytest
import torch

from gamer_amd import data as gdata
from gamer_amd import synthetic


@pytest.fixture(scope="module")
def fx(tmp_path_factory):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_small.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta_json"]))
    root = tmp_path_factory.mktemp("smbdata")
    d = os.path.join(root, meta["name"])
    os.makedirs(d)
    for suffix, text in meta["files"].items():
        with open(os.path.join(d, meta["name"] + suffix), "w") as f:
            f.write(text)
    ds = gdata.SMBData(str(root), meta["name"])
    return z, meta, ds


def test_vocabulary_matches_the_reference_tokenizer(fx):
    z, meta, ds = fx
    assert [str(t) for t in z["new_tokens"]] == ds.new_tokens
    assert [ds.tokens[t] for t in ds.new_tokens] == z["new_token_ids"].tolist()
    assert len(ds.tokens) == int(z["vocab_size"]) and ds.tokens.pad_id == int(z["pad_id"])
    assert [str(b) for b in z["behaviors"]] == ds.behaviors
    assert [ds.behavior_token_ids[b] for b in ds.behaviors] == z["behavior_token_ids"].tolist()
    assert ds.target_behavior == str(z["target_behavior"])
    # add_tokens() numbers the sorted list: '<a_10>' sorts before '<a_2>', '<b_*>' before '<behavior_*>' before '<c_*>'
    assert ds.tokens["<a_10>"] < ds.tokens["<a_2>"]
    assert max(ds.tokens[t] for t in ds.new_tokens if t.startswith("<b_")) < min(ds.behavior_token_ids.values())


@pytest.mark.parametrize("mode", ["train", "valid"])
def test_training_and_validation_batches_bit_exact(fx, mode):
    z, meta, ds = fx
    ss = ds.train_samples(meta["max_his_len"], augment=meta["augment"]) if mode == "train" else ds.valid_samples(meta["max_his_len"])
    assert len(ss) == int(z[f"n_{mode}"])
    coll = gdata.Collator(ds)
    nb = int(z[f"n_{mode}_batches"])
    chunks = gdata.batches(len(ss), meta["batch"])
    assert len(chunks) == nb
    for b, idx in enumerate(chunks):
        got = coll.train(ss, idx)
        for k in ("input_ids", "attention_mask", "labels", "session_ids", "extended_session_ids", "actions"):
            ref = z[f"{mode}/{b}/{k}"]
            assert got[k].dtype == torch.int64 and tuple(got[k].shape) == ref.shape, (k, b)
            assert np.array_equal(got[k].numpy(), ref), (mode, b, k)
        assert got["time"].dtype == torch.float32
        assert np.array_equal(got["time"].numpy(), z[f"{mode}/{b}/time"]), (mode, b)
        assert got["behavior"] == [str(x) for x in z[f"{mode}/{b}/behavior"]]
        assert got["split"] == mode
    if mode == "train":
        # the fixture exercises cropping to max_his_len and the augmentation (more samples than users)
        assert len(ss) > len({u for u in ss.users})
        assert int((ss.ptr[1:] - ss.ptr[:-1]).max()) == (meta["max_his_len"] + 1) * ds.token_count


def test_per_interaction_training_batches_bit_exact(fx):
    """tasks=smb_explicit: SMBExplicitDataset(mode="train") + DecoderOnlyCollator(only_train_response=True); the
    first session's targets have an empty history."""
    z, meta, ds = fx
    ss = ds.train_samples_per_interaction(meta["max_his_len"])
    assert len(ss) == int(z["n_train_per_interaction"])
    assert int((ss.n_history == 0).sum()) > 0
    coll = gdata.Collator(ds)
    chunks = gdata.batches(len(ss), 4 * meta["batch"])
    assert len(chunks) == int(z["n_train_pi_batches"])
    for b, idx in enumerate(chunks):
        got = coll.train(ss, idx, only_train_response=True)
        for k in ("input_ids", "attention_mask", "labels", "session_ids", "extended_session_ids", "actions"):
            assert np.array_equal(got[k].numpy(), z[f"train_pi/{b}/{k}"]), (b, k)
        assert np.array_equal(got["time"].numpy(), z[f"train_pi/{b}/time"]), b


def test_test_batches_targets_and_candidates_bit_exact(fx):
    z, meta, ds = fx
    full = ds.test_samples(meta["max_his_len"])
    assert len(full) == int(z["n_test"])
    coll = gdata.Collator(ds)
    for beh in ds.behaviors:
        sub = full.filter_by_behavior(beh)
        assert len(sub) == int(z[f"test/{beh}/n"])
        for b, idx in enumerate(gdata.batches(len(sub), meta["batch"])):
            inputs, targets = coll.test(sub, idx)
            for k in ("input_ids", "attention_mask", "session_ids", "extended_session_ids", "actions"):
                assert np.array_equal(inputs[k].numpy(), z[f"test/{beh}/{b}/{k}"]), (beh, b, k)
            ref_t = json.loads(str(z[f"test/{beh}/{b}/target_ids_json"]))
            assert len(targets) == len(ref_t)
            for tg, rt in zip(targets, ref_t):
                # the reference's target strings carry the behaviour token in front of the item tokens
                assert [[ds.behavior_token_ids[beh]] + row for row in tg.tolist()] == rt
            # completed prompt (test_SMB_decoder.py:107-121): behaviour token, mask 1, level appended
            inputs2, _ = coll.test(sub, idx, behavior=beh)
            assert torch.equal(inputs2["input_ids"][:, :-1], inputs["input_ids"])
            assert bool((inputs2["input_ids"][:, -1] == ds.behavior_token_ids[beh]).all())
            assert bool((inputs2["attention_mask"][:, -1] == 1).all())
            assert bool((inputs2["actions"][:, -1] == ds.behavior_level[beh]).all())
            assert inputs2["session_ids"].shape == inputs2["input_ids"].shape
        assert np.array_equal(ds.candidate_tokens(beh), z[f"test/{beh}/candidates"])


def test_sample_layout_feeds_the_model_contract(fx):
    """What gamer_amd.engine / the router expect: item-aligned sequences of 5 tokens, behaviour
    token first, labels never on a behaviour token or padding, actions padded with 100, session ids non-decreasing
    along the kept tokens (what gamer_session_spans requires)."""
    z, meta, ds = fx
    ss = ds.train_samples(meta["max_his_len"], augment=meta["augment"])
    got = gdata.Collator(ds).train(ss, np.arange(len(ss)))
    ids, am, lab, act, sid = (got[k] for k in ("input_ids", "attention_mask", "labels", "actions", "session_ids"))
    assert ids.shape[1] % ds.token_count == 0
    beh_ids = torch.tensor(sorted(ds.behavior_token_ids.values()))
    first = ids[:, ::ds.token_count]
    assert bool((torch.isin(first, beh_ids) | (first == ds.tokens.pad_id)).all())
    assert bool((lab[:, ::ds.token_count] == -100).all()) and bool((lab[am == 0] == -100).all())
    assert bool((act[am == 0] == 100).all()) and bool((act[am == 1] <= ds.max_behavior_level).all())
    kept_sid = torch.where(am.bool(), sid, torch.full_like(sid, 10 ** 9))
    assert bool((kept_sid[:, 1:] >= kept_sid[:, :-1]).all())


def test_writer_round_trip_and_sharding(tmp_path):
    d = synthetic.write_smb_dataset(str(tmp_path), "T", n_users=12, n_items=30, codebook=8, seed=5)
    assert os.path.isdir(d)
    ds = gdata.SMBData(str(tmp_path), "T")
    ss = ds.train_samples(20, augment=4)
    assert len(ss) >= 1
    a = gdata.batches(len(ss), 4, shuffle_seed=1, rank=0, world=2)
    b = gdata.batches(len(ss), 4, shuffle_seed=1, rank=1, world=2)
    allidx = np.sort(np.concatenate(a + b))
    assert np.array_equal(allidx, np.arange(len(ss)))
    with pytest.raises(ValueError):
        ds.train_samples(20, augment=0)
