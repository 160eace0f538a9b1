"""Resident-K/V attention kernels (csrc/attention_res.hip; attn_*_br_kernel in csrc/attention_bf16.hip) in the regimes the shape grid of
tests/test_ops_gpu.py does not reach: several key blocks (S up to 600: three blocks of 256 keys, five query blocks of 128 slots), a few
persistent workgroups walking many (sequence, kv head) units one after the other (LDS state, queue counters and carried partial sums
across units), both unit forms (whole pair / one head - one key parity - per workgroup), dropout-mask consistency with the tiled kernels.
Reference: a dense fp64 attention (the oracle's masks), at the bars of the main grid; and the tiled kernels on the same inputs."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

from gamer_amd import ops  # noqa: E402


def _helpers():
    import test_ops_gpu as T
    return T


_env = ops.env_switches          # (sets the switches and has the library re-read them: they are cached per process)


@pytest.mark.parametrize("grid,split", [(0, None), (3, 0), (3, 1), (5, None)])
@pytest.mark.parametrize("cross,use_order", [(False, False), (True, True), (True, False)])
@pytest.mark.parametrize("n_items,B", [(120, 4), (60, 7), (101, 5)])
def test_resident_kernels_over_blocks_units_and_few_workgroups(n_items, B, cross, use_order, grid, split):
    """S = 600 / 300 / 505 (three, two, two key blocks), G = 2; grid = 3 or 5 persistent workgroups for 12-21 pairs (every workgroup
    walks several units), split = the unit form forced either way."""
    T = _helpers()
    from gamer_amd import synthetic
    from oracle import qwen3multi_oracle as orc
    nq, nkv = 6, 3
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=3 + n_items, pad_rows={0: max(1, n_items // 3)})
    S = batch["input_ids"].shape[1]
    g = torch.Generator().manual_seed(n_items + B)
    q, k, v = (torch.randn(B, S, h, 64, generator=g) for h in (nq, nkv, nkv))
    d_o = torch.randn(B, S, nq, 64, generator=g)
    self_ok, cross_ok = orc.mask_predicates(batch["attention_mask"], batch["actions"])
    ok = cross_ok if cross else self_ok
    leaves = [t.double().requires_grad_(True) for t in (q, k, v)]
    o_ref, lse_ref, empty = T._attn_ref(*leaves, ok, nq, nkv, 0.125)
    (o_ref * d_o.double()).sum().backward()
    env = {"GAMER_ATTN_RES": 1}
    if grid:
        env["GAMER_ATTN_RES_GRID"] = grid
    if split is not None:
        env["GAMER_ATTN_RES_SPLIT"] = split
    with _env(**env):
        res = T._run_attn(batch, cross, B, S, nq, nkv, q=q, k=k, v=v, d_o=d_o, use_order=use_order, spill="split_h2")
        torch.cuda.synchronize()
    Tn = B * S
    ne = ~empty
    assert T._rel(res["o"], o_ref.reshape(Tn, -1)) < 2e-5
    assert float((res["lse"].cpu().permute(0, 2, 1)[ne].double() - lse_ref.permute(0, 2, 1)[ne]).abs().max()) < 2e-5
    assert T._rel(res["dq"], leaves[0].grad.reshape(Tn, -1)) < 5e-5
    assert T._rel(res["dk"], leaves[1].grad.reshape(Tn, -1)) < 5e-5
    assert T._rel(res["dv"], leaves[2].grad.reshape(Tn, -1)) < 5e-5


@pytest.mark.parametrize("cross", [False, True])
def test_resident_and_tiled_kernels_draw_the_same_dropout_mask(cross):
    """Same seed, dropout 0.3: the resident kernels and the tiled ones regenerate the same keep mask, so their outputs and gradients agree
    to rounding (the mask function is shared by every attention kernel of the library)."""
    T = _helpers()
    from gamer_amd import synthetic
    B, n_items, nq, nkv = 6, 101, 6, 3
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=21, pad_rows={1: 40})
    S = batch["input_ids"].shape[1]
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, S, h, 64, generator=g) for h in (nq, nkv, nkv))
    d_o = torch.randn(B, S, nq, 64, generator=g)
    out = {}
    for form in (0, 1):
        with _env(GAMER_ATTN_RES=form, GAMER_ATTN_RES_GRID=4):
            out[form] = T._run_attn(batch, cross, B, S, nq, nkv, p_drop=0.3, seed=99, q=q, k=k, v=v, d_o=d_o, use_order=cross,
                                    spill="split_h2")
            torch.cuda.synchronize()
    for key in ("o", "dq", "dk", "dv"):
        a, b_ = out[0][key].double().cpu(), out[1][key].double().cpu()
        assert float((a - b_).abs().max()) < 5e-6 * float(a.abs().max()), key


@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("p_drop", [0.0, 0.25])
def test_bf16_resident_forward_and_dq_are_bit_identical_to_the_tiled_kernels(cross, p_drop):
    """(and the head-resident dK / dV kernel against the tiled one, to rounding)"""
    from gamer_amd import ops, synthetic
    from gamer_amd.config import synthetic_config
    B, items, nq, nkv = 9, 101, 6, 3
    cfg = synthetic_config()
    S = items * 5
    Tn = B * S
    dev = "cuda"
    batch = synthetic.make_batch(B, items, 256, 3, seed=3, pad_rows={2: 30}, behavior_probs=[0.7, 0.25, 0.05])
    r = ops.alloc_router_outputs(B, S, dev)
    ops.router_fwd(batch["input_ids"].to(dev), batch["attention_mask"].to(dev), batch["actions"].to(dev), cfg.behavior_lut().to(dev), 5, 4, 8, r)
    torch.manual_seed(1)
    bf = torch.bfloat16
    q = torch.randn(Tn, nq * 64, device=dev).to(bf)
    k = torch.randn(Tn, nkv * 64, device=dev).to(bf)
    qkv = torch.randn(Tn, (nq + 2 * nkv) * 64, device=dev).to(bf)
    v = qkv[:, (nq + nkv) * 64:]
    do = torch.randn(Tn, nq * 64, device=dev).to(bf)
    n_t = (S + 31) // 32
    perm = torch.empty(B, S, dtype=torch.int32, device=dev)
    tk = torch.empty(B, n_t, dtype=torch.int32, device=dev)
    tm = torch.empty(B, n_t, dtype=torch.int32, device=dev)
    ops.attn_row_order(r["empty_cross"], perm, tk, tm)
    kl, ql, od = (r["kl_cross"], r["ql_cross"], (perm, tm, r["empty_cross"])) if cross else (r["kl_self"], None, None)
    out = {}
    for form, grid in ((0, 0), (1, 4)):
        env = {"GAMER_ATTN_RES": form}
        if grid:
            env["GAMER_ATTN_RES_GRID"] = grid
        with _env(**env):
            o = torch.full((Tn, nq * 64), float("nan"), device=dev, dtype=bf)
            lse = torch.full((B, nq, S), float("nan"), device=dev)
            ops.attn_fwd_bf16(q, nq * 64, k, nkv * 64, v, qkv.shape[1], kl, ql, B, S, nq, nkv, 0.125, p_drop, 7, o, lse, order=od)
            delta = torch.zeros(B, nq, S, device=dev)
            dq = torch.full((Tn, nq * 64), float("nan"), device=dev, dtype=bf)
            dk = torch.full((Tn, nkv * 64), float("nan"), device=dev, dtype=bf)
            dqkv = torch.zeros_like(qkv)
            dv = dqkv[:, (nq + nkv) * 64:]
            ops.attn_bwd_bf16(q, nq * 64, k, nkv * 64, v, qkv.shape[1], o, do, lse, kl, ql, B, S, nq, nkv, 0.125, p_drop, 7, delta, dq,
                              nq * 64, dk, nkv * 64, dv, qkv.shape[1], order=od)
            torch.cuda.synchronize()
            out[form] = dict(o=o, lse=lse, dq=dq, dk=dk, dv=dv.clone())
    for key in ("o", "lse", "dq"):
        assert torch.equal(out[0][key], out[1][key]), key
    # dK / dV: the head-resident kernel (attn_bwd_dkv_bh_kernel) adds the heads of a kv group and the query tiles in another order than
    # the tiled one - same products, fp32 sums, one rounding to bf16 at the end: the two agree to an ulp of bf16 on the tensor's scale
    for key in ("dk", "dv"):
        a, b_ = out[0][key].float(), out[1][key].float()
        assert float((a - b_).abs().max()) <= 2.0 ** -7 * float(a.abs().max()), key
