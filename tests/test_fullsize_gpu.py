"""Parity at BASELINE.json's full size (per-GPU batch 1024 x 505 tokens, shipped dims), where the CPU oracle cannot run the
whole batch: size-independent properties of the train step, tied to the oracle on rows sampled from the full batch.

  (a) sequences do not interact: the logits of EVERY row of the 1024-row batch equal the logits of the same row run inside
      a 128-row batch (all 1024 rows, eight chunks), and for twelve rows spread over the batch also the logits of those
      rows run alone and the CPU oracle's logits for them (north-star bar: 1e-3 of the abs-max);
  (b) the loss is a normalised sum: with num_items_in_batch fixed, loss(full) = loss(first half) + loss(second half);
  (c) the gradient is linear in the batch: the gradient of the full batch equals the accumulated gradient of its two halves
      (fp32 atomics in the split-K weight gradients: summation order differs, tolerance 2e-4 of each tensor's abs-max),
      and the oracle's gradient of a SMALL sub-batch equals the engine's on that sub-batch (ties the backward to the oracle
      in the same process / same weights as the full-size run);
  (d) dropout masks are a function of (seed, step, position) only: a training forward with dropout repeated with the same
      step counter gives the same loss bit for bit.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import synthetic  # noqa: E402
from gamer_amd.config import synthetic_config  # noqa: E402
from gamer_amd.engine import Engine  # noqa: E402
from oracle import qwen3multi_oracle as orc  # noqa: E402

B, ITEMS = 1024, 101
ROWS = [0, 1, 2, 127, 128, 255, 511, 512, 517, 768, 1022, 1023]     # rows compared against the CPU oracle


def _sub(batch, idx):
    return {k: v[idx].contiguous() for k, v in batch.items()}


def _relmax(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("dtype,matmul", [("f32", "f32"), ("f32", "split3"), ("f32", "split6"), ("bf16", "f32")])
def test_full_batch_properties_and_oracle_rows(dtype, matmul):
    cfg = synthetic_config()
    ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
    sd = orc.init_state_dict(ocfg, seed=5)
    eng = Engine(cfg, temperature=0.7, dtype=dtype, matmul=matmul)
    eng.load_state_dict(sd)
    amp = dtype == "bf16"
    tol_logits = 1e-2 if amp else 1e-3       # the bars of tests/test_bf16_gpu.py / the north star
    batch = synthetic.make_batch(B, ITEMS, 256, 3, seed=20251114, behavior_probs=[0.7, 0.25, 0.05])
    n_items = float((batch["labels"][:, 1:] != -100).sum())
    idx = torch.tensor(ROWS)

    # (a) rows of the full batch == the same rows alone == the oracle
    loss_full, logits = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"],
                                    num_items_in_batch=n_items, train=False)
    loss_full = float(loss_full)
    rows_full = logits[idx.to(logits.device)].float().cpu().numpy()
    full = logits.clone()                    # (the returned logits are a view of the engine's workspace)
    small = _sub(batch, idx)
    _, lg_small = eng.forward(small["input_ids"], small["attention_mask"], small["actions"], labels=small["labels"],
                              num_items_in_batch=n_items, train=False)
    e_small = _relmax(lg_small.float().cpu().numpy(), rows_full)
    out = orc.forward(sd, ocfg, small["input_ids"], small["attention_mask"], small["actions"], labels=small["labels"],
                      temperature=0.7, num_items_in_batch=n_items, amp=amp)
    e_oracle = _relmax(rows_full, out["logits"].float().detach().numpy())
    assert e_small < (5e-3 if amp else 1e-5), e_small
    assert e_oracle < tol_logits, e_oracle
    # ... and every row of the batch against the same row inside a 128-row batch (different tile / work-list positions)
    scale = float(full.float().abs().max())
    worst_chunk = 0.0
    for c0 in range(0, B, 128):
        cb = _sub(batch, torch.arange(c0, c0 + 128))
        _, lg_c = eng.forward(cb["input_ids"], cb["attention_mask"], cb["actions"], labels=cb["labels"],
                              num_items_in_batch=n_items, train=False)
        worst_chunk = max(worst_chunk, float((lg_c.float() - full[c0:c0 + 128].float()).abs().max()) / scale)
    del full
    assert worst_chunk < (5e-3 if amp else 1e-5), worst_chunk

    # (b) the loss is a normalised sum over the rows
    halves = [_sub(batch, torch.arange(0, B // 2)), _sub(batch, torch.arange(B // 2, B))]
    eng.zero_grad()
    parts = []
    for hb in halves:
        l, _ = eng.forward(hb["input_ids"], hb["attention_mask"], hb["actions"], labels=hb["labels"],
                           num_items_in_batch=n_items, train=True, dropout=False)
        parts.append(float(l))
        eng.backward(1.0)
    assert abs(sum(parts) - loss_full) < (2e-3 if amp else 2e-6) * loss_full, (parts, loss_full)
    g_halves = {k: g.clone() for k, g in eng.grads.items()}

    # (c) gradient of the full batch == accumulated gradient of the halves
    eng.zero_grad()
    eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"],
                num_items_in_batch=n_items, train=True, dropout=False)
    eng.backward(1.0)
    worst, wk = 0.0, None
    for k, g in eng.grads.items():
        den = float(g.abs().max())
        if den == 0.0:
            assert float(g_halves[k].abs().max()) == 0.0
            continue
        e = float((g - g_halves[k]).abs().max()) / den
        if e > worst:
            worst, wk = e, k
    assert worst < (3e-2 if amp else 2e-4), (wk, worst)
    assert all(bool(torch.isfinite(g).all()) for g in eng.grads.values())
    # ... and the oracle's gradient on the small sub-batch, same engine, same weights
    _, g_ref, _ = orc.loss_and_grads(sd, ocfg, small, temperature=0.7, amp=amp)
    eng.zero_grad()
    eng.forward(small["input_ids"], small["attention_mask"], small["actions"], labels=small["labels"], train=True, dropout=False)
    eng.backward(1.0)
    worst_o = max(_relmax(eng.grads[k].cpu().numpy(), g.numpy()) for k, g in g_ref.items())
    assert worst_o < (8e-2 if amp else 1e-3), worst_o

    # (d) dropout masks depend on (seed, step, position) only
    step = eng.dropout_step
    l1, _ = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"], train=True)
    eng.dropout_step = step
    l2, _ = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"], train=True)
    assert float(l1) == float(l2)
    assert abs(float(l1) - loss_full * n_items / n_items) < 0.5        # dropout moves the loss, not its scale
    eng.check_inputs()
