"""BASELINE configs[3]: Qwen3Multi + 4x generative augmentation (``tasks=smb_explicit_decoder_4``), per-GPU micro-batch 128 x
gradient accumulation 4 (README recipe, ref:README.md:63), at the shipped dims - through the REAL data path: a dataset
directory in the reference's on-disk format -> ``SMBData.train_samples(100, augment=4)`` (the thinning of
ref:SeqRec/datasets/SMB_dataset.py:540-584) -> ``Collator.train`` (right padding to each batch's longest sample) ->
``Engine.train_window`` (HF Trainer's accumulation window: every micro-batch's loss is sum CE / the WINDOW's label count,
gradients accumulate, one clip + AdamW at the end).

Checked: the batches really are ragged and differ in length; the window's gradient is the sum of its four micro-batches'
(linearity; fp32 atomics in the split-K weight gradients); the gradient of a sub-batch equals the CPU oracle's on the same
rows and weights (the north star's 1e-3, measured ~1e-5); ``train_window`` equals the hand-written loop; the bf16 window
runs the same shapes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import data as gdata, synthetic  # noqa: E402
from gamer_amd.engine import Engine  # noqa: E402
from oracle import qwen3multi_oracle as orc  # noqa: E402

KEYS = ("input_ids", "attention_mask", "actions", "labels")
B_DEV, ACCUM, HIS = 128, 4, 100


@pytest.fixture(scope="module")
def window(tmp_path_factory):
    root = tmp_path_factory.mktemp("smb")
    synthetic.write_smb_dataset(str(root), "syn", n_users=150, n_items=900, codebook=256, max_sessions=45, max_per_session=4,
                                min_sessions=6, seed=3)
    ds = gdata.SMBData(str(root), "syn")
    samples = ds.train_samples(HIS, augment=4)
    assert len(samples) > 3 * 150                      # the augmentation added copies
    coll = gdata.Collator(ds)
    order = np.random.RandomState(0).permutation(len(samples))[:B_DEV * ACCUM]
    micro = [coll.train(samples, order[a * B_DEV:(a + 1) * B_DEV]) for a in range(ACCUM)]
    cfg = gdata.model_config(ds, HIS)
    return cfg, micro


def _grad_of(eng, batches, n_items):
    eng.zero_grad()
    losses = []
    for b in batches:
        loss, _ = eng.forward(b["input_ids"], b["attention_mask"], b["actions"], labels=b["labels"], num_items_in_batch=n_items,
                              train=True, dropout=False)
        eng.backward(1.0)
        losses.append(float(loss))
    return eng.flat_g.clone(), losses


def test_window_shapes_are_ragged(window):
    cfg, micro = window
    assert cfg.hidden_size == 256 and cfg.num_hidden_layers == 8 and cfg.num_attention_heads == 6 and cfg.vocab_size > 900
    S = [b["input_ids"].shape[1] for b in micro]
    assert max(S) == (HIS + 1) * 5                      # the longest history is cropped to max_his_len
    lens = torch.cat([b["attention_mask"].sum(1) for b in micro])
    assert int(lens.min()) < 0.5 * int(lens.max()) and len(set(lens.tolist())) > 20          # ragged
    assert all(b["input_ids"].shape[0] == B_DEV for b in micro)


@pytest.mark.parametrize("matmul", ["split3", "f32"])
def test_accumulated_gradient_is_the_sum_and_matches_the_oracle(window, matmul):
    cfg, micro = window
    ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
    sd = orc.init_state_dict(ocfg, seed=9)
    eng = Engine(cfg, temperature=0.7, matmul=matmul)
    eng.load_state_dict(sd)
    n_items = float(sum(int((b["labels"][:, 1:] != -100).sum()) for b in micro))
    g_win, l_win = _grad_of(eng, micro, n_items)
    g_sum = torch.zeros_like(g_win)
    for a, b in enumerate(micro):
        g_a, l_a = _grad_of(eng, [b], n_items)
        g_sum += g_a
        assert abs(l_a[0] - l_win[a]) <= 1e-6 * abs(l_win[a])
    views_w, views_s = eng.layout.views(g_win), eng.layout.views(g_sum)
    for k in views_w:
        d = float((views_w[k] - views_s[k]).abs().max())
        assert d <= 2e-5 * float(views_w[k].abs().max()) + 1e-12, (k, d)
    # a sub-batch of the first micro-batch against the CPU oracle (same weights, same normaliser)
    rows = torch.tensor([0, 17, 64, 127])
    sub = {k: micro[0][k][rows].contiguous() for k in KEYS}
    assert len(set(sub["attention_mask"].sum(1).tolist())) > 1
    g_sub, l_sub = _grad_of(eng, [sub], n_items)
    ref_loss, ref_grads, _ = orc.loss_and_grads(sd, ocfg, sub, temperature=0.7, num_items_in_batch=n_items)
    assert abs(l_sub[0] - float(ref_loss)) <= 2e-6 * abs(float(ref_loss)) + 1e-9
    got = eng.layout.views(g_sub)
    worst = max(float((got[k].cpu() - g).abs().max() / g.abs().max().clamp_min(1e-30)) for k, g in ref_grads.items())
    assert worst < 1e-3, worst                          # north-star bar; measured ~1e-5


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_window_is_the_hand_written_loop(window, dtype):
    cfg, micro = window
    sd = orc.init_state_dict(orc.OracleConfig.from_dict(cfg.to_dict()), seed=9)
    a, b = Engine(cfg, temperature=0.7, dtype=dtype), Engine(cfg, temperature=0.7, dtype=dtype)
    for e in (a, b):
        e.load_state_dict(sd)
        e.base_seed = 77                                # same dropout masks on both sides (dropout is ON: the recipe's step)
    losses = a.train_window(micro, 5e-4)
    n_items = float(sum(int((m["labels"][:, 1:] != -100).sum()) for m in micro))
    b.zero_grad()
    ref_losses = []
    for m in micro:
        loss, _ = b.forward(m["input_ids"], m["attention_mask"], m["actions"], labels=m["labels"], num_items_in_batch=n_items,
                            train=True)
        b.backward(1.0)
        ref_losses.append(float(loss))
    b.optimizer_step(5e-4)
    tol = 5e-3 if dtype == "bf16" else 2e-5
    for x, y in zip(losses, ref_losses):
        assert abs(float(x) - y) <= 1e-6 * abs(y)
    assert abs(float(a.grad_norm) - float(b.grad_norm)) <= tol * float(b.grad_norm)
    assert bool(torch.isfinite(a.flat_p).all())
    # same update (Adam's first step is +-lr per element: compare where the gradient is well away from zero)
    big = b.flat_g.abs() > 1e-3 * b.flat_g.abs().max()
    assert float((a.flat_p - b.flat_p)[big].abs().max()) <= (2e-2 if dtype == "bf16" else 1e-3) * 5e-4
