"""The injecting layers' behaviour-embedding columns as a table (csrc/inject.hip, Engine.split_inject): the three kernels against
fp64 torch, the table forms of the SwiGLU kernels against the plain ones on pre-added inputs, and the engine with the split against
the engine on the reference's concatenated [T, 320] expert input (ref:SeqRec/models/generative/Qwen3Moe/FFN.py:53-72) and the oracle."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import ops, synthetic  # noqa: E402
from gamer_amd.config import synthetic_config  # noqa: E402
from gamer_amd.engine import Engine  # noqa: E402
from oracle import qwen3multi_oracle as orc  # noqa: E402

DEV = "cuda"


def _segments(T, nseg, seed):
    g = torch.Generator().manual_seed(seed)
    cuts = torch.sort(torch.randint(0, T + 1, (nseg - 1,), generator=g)).values
    cuts[nseg // 2] = cuts[nseg // 2 - 1]                      # an empty segment
    return torch.cat([torch.zeros(1, dtype=torch.long), cuts, torch.tensor([T])]).to(torch.int32)


@pytest.mark.parametrize("E,NB1,twoI,EB,din", [(6, 4, 1024, 64, 320), (3, 5, 96, 8, 40)])
def test_inject_table_kernels_match_fp64(E, NB1, twoI, EB, din):
    torch.manual_seed(E + NB1)
    H = din - EB
    Eb = torch.randn(NB1, EB, device=DEV)
    W = torch.randn(E * twoI, din, device=DEV) * 0.05
    tbl = torch.empty(E * NB1, twoI, device=DEV)
    ops.inject_table_fwd(Eb, W, din, H, E, twoI, tbl)
    ref = torch.einsum("bj,enj->ebn", Eb.double(), W.double().view(E, twoI, din)[:, :, H:]).reshape(E * NB1, twoI)
    assert float((tbl.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())

    T, nseg = 5000, E * NB1
    offs = _segments(T, nseg, 3).to(DEV)
    x = torch.randn(T, twoI + 8, device=DEV)[:, :twoI]        # (a leading dimension larger than the width)
    ws = torch.empty(ops.segment_colsum_ws_floats(T, twoI, nseg), device=DEV)
    seg = torch.full((nseg, twoI), float("nan"), device=DEV)
    ops.segment_colsum(x, x.stride(0), T, twoI, offs, nseg, ws, seg)
    o = offs.cpu().tolist()
    ref_seg = torch.stack([x[o[s]:o[s + 1]].double().sum(0) for s in range(nseg)])
    assert float((seg.double() - ref_seg).abs().max()) < 2e-5 * float(ref_seg.abs().max())
    seg2 = torch.empty_like(seg)
    ops.segment_colsum(x, x.stride(0), T, twoI, offs, nseg, ws, seg2)
    assert torch.equal(seg, seg2)                               # a fixed summation order

    dW = torch.randn(E * twoI, din, device=DEV)
    dEb = torch.randn(NB1, EB, device=DEV)
    dW0, dEb0 = dW.clone(), dEb.clone()
    ops.inject_table_bwd(seg, Eb, W, din, H, E, twoI, dW, dEb, torch.empty(NB1 * E * EB, device=DEV))
    s3 = ref_seg.view(E, NB1, twoI)
    ref_dW = dW0.double().clone().view(E, twoI, din)
    ref_dW[:, :, H:] += torch.einsum("ebn,bj->enj", s3, Eb.double())
    ref_dEb = dEb0.double() + torch.einsum("ebn,enj->bj", s3, W.double().view(E, twoI, din)[:, :, H:])
    assert torch.equal(dW[:, :H], dW0[:, :H])                   # the hidden columns are not touched
    assert float((dW.double().view(E, twoI, din) - ref_dW).abs().max()) < 2e-5 * float(ref_dW.abs().max())
    assert float((dEb.double() - ref_dEb).abs().max()) < 2e-5 * float(ref_dEb.abs().max())


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_swiglu_table_forms_equal_the_plain_forms_on_the_sums(p):
    torch.manual_seed(1)
    T, I, G = 777, 512, 24
    gu = torch.randn(T, 2 * I, device=DEV)
    tbl = torch.randn(G, 2 * I, device=DEV)
    rg = torch.randint(0, G, (T,), device=DEV, dtype=torch.int32)
    summed = gu + tbl[rg.long()]
    hm_ref, hm = torch.empty(T, I, device=DEV), torch.empty(T, I, device=DEV)
    ops.swiglu_fwd_ld(summed, 2 * I, T, I, p, 11, hm_ref)
    ops.swiglu_fwd_ld_tbl(gu, 2 * I, T, I, p, 11, hm, tbl, rg)
    assert torch.equal(hm, hm_ref)
    dhm = torch.randn(T, I, device=DEV)
    a, b = summed.clone(), gu.clone()
    ops.swiglu_bwd_ld(a, 2 * I, T, I, dhm, p, 11)
    ops.swiglu_bwd_ld_tbl(b, 2 * I, T, I, dhm, p, 11, tbl, rg)
    assert torch.equal(a, b)


def _grads(split, matmul, batch, sd, cfg, fast_kernels=False, dropout=False):
    os.environ["GAMER_SPLIT_INJECT"] = "1" if split else "0"
    try:
        eng = Engine(cfg, temperature=0.7, matmul=matmul)
    finally:
        os.environ.pop("GAMER_SPLIT_INJECT", None)
    assert eng.split_inject == split
    eng.load_state_dict(sd)
    eng.base_seed = 5
    env = dict(GAMER_GEMM_AS_MIN_M=1, GAMER_GEMM_OS_MIN_M=1) if fast_kernels else {}
    with ops.env_switches(**env):
        for _ in range(2):                                      # (second pass: the packed weight pieces exist)
            eng.dropout_step = 0
            loss, logits = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"],
                                       train=True, dropout=dropout)
            logits = logits.clone()
            eng.zero_grad()
            eng.backward(1.0)
        torch.cuda.synchronize()
    return float(loss), logits, {k: v.clone() for k, v in eng.grads.items()}


@pytest.mark.parametrize("matmul,fast", [("split3", False), ("split3", True), ("f32", False), ("split6", False)])
def test_engine_with_the_table_split_matches_the_concatenated_input_and_the_oracle(matmul, fast):
    cfg = synthetic_config()
    ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
    sd = orc.init_state_dict(ocfg, seed=4)
    batch = synthetic.make_batch(6, 40, 256, 3, ragged=True, seed=9, behavior_probs=[0.7, 0.25, 0.05])
    l0, lg0, g0 = _grads(False, matmul, batch, sd, cfg)
    l1, lg1, g1 = _grads(True, matmul, batch, sd, cfg, fast_kernels=fast)
    assert abs(l0 - l1) < 2e-6 * abs(l0)
    assert float((lg0 - lg1).abs().max()) < 2e-5 * float(lg0.abs().max())
    for k in g0:
        d = float((g0[k] - g1[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-20)
        assert d < 2e-4, (k, d)
    ref_loss, ref_grads, _ = orc.loss_and_grads(sd, ocfg, batch, temperature=0.7)
    assert abs(l1 - float(ref_loss)) < 1e-5 * float(ref_loss)
    worst = max(float((g1[k].cpu() - g).abs().max() / g.abs().max().clamp_min(1e-20)) for k, g in ref_grads.items())
    assert worst < 1e-3, worst
    beh_keys = [k for k in ref_grads if "behavior_embedding" in k and "mlp" in k]
    assert beh_keys and all(float(g1[k].abs().max()) > 0 for k in beh_keys)


def test_table_split_with_dropout_is_reproducible():
    """Dropout on: the SwiGLU mask is a function of the SORTED row index, and the split sorts the expert rows by (expert, behaviour)
    instead of by expert alone - another (equally valid) assignment of masks to tokens, so the two forms are not comparable number
    by number.  The split engine repeats its own loss and gradient bit for bit, and its loss stays near the other form's."""
    cfg = synthetic_config()
    sd = orc.init_state_dict(orc.OracleConfig.from_dict(cfg.to_dict()), seed=4)
    batch = synthetic.make_batch(6, 40, 256, 3, ragged=True, seed=9, behavior_probs=[0.7, 0.25, 0.05])
    l0, _, _ = _grads(False, "split3", batch, sd, cfg, dropout=True)
    l1, _, g1 = _grads(True, "split3", batch, sd, cfg, dropout=True)
    l2, _, g2 = _grads(True, "split3", batch, sd, cfg, dropout=True)
    assert l1 == l2 and abs(l0 - l1) < 0.05 * abs(l0)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
        assert bool(torch.isfinite(g1[k]).all()), k
