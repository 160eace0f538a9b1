"""Per-kernel tests of the bf16 (AMP) entry points, through the C ABI on the GPU.

GEMMs: small-integer operands are exact in bf16 and their dot products exact in fp32, so the kernel must reproduce an
fp64 product bit for bit (after the one rounding to bf16 where the output is bf16) - any wrong fragment lane, k order,
swizzle or edge shows up as a hard mismatch, not as noise.  Elementwise twins: the same fp32 arithmetic as the fp32 entry
point on bf16-representable inputs, rounded once.  Attention: dense fp64 reference with the bf16 run's rule for query
rows without an allowed key (output 0, no gradient - see oracle.qwen3multi_oracle.attention), bf16-level tolerances.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import ops, synthetic  # noqa: E402
from oracle import qwen3multi_oracle as orc  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16


def dev(t):
    return t.to(DEV).contiguous()


def ints(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=g).float()


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,ldc", [(128, 128, 64, 128), (256, 384, 256, 384), (300, 200, 128, 200), (517, 1041, 256, 1088),
                                       (1024, 320, 512, 320), (96, 64, 320, 64)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_bf16_forward_exact(M, N, K, ldc, accumulate):
    g = torch.Generator().manual_seed(M + N + K)
    A, B = ints((M, K), g), ints((N, K), g)
    B = B + torch.arange(N)[:, None] % 3                       # asymmetric: a row/column swap cannot cancel
    C0 = ints((M, ldc), g, -2, 3)
    a16, b16 = dev(A.to(BF)), dev(B.to(BF))
    c = dev(C0.to(BF))
    ops.gemm(a16, K, 1, b16, K, 1, c, ldc, M, N, K, accumulate=accumulate)
    ref = A.double() @ B.double().T + (C0[:, :N].double() if accumulate else 0)
    assert torch.equal(c.cpu()[:, :N].float(), ref.to(BF).float())
    assert torch.equal(c.cpu()[:, N:].float(), C0[:, N:].to(BF).float()), "columns past N must stay untouched"


@pytest.mark.parametrize("M,N,K,ldc", [(4096, 256, 64, 256), (4196, 512, 256, 520), (8192 + 77, 768, 128, 768),
                                       (4224 + 50, 320, 512, 320), (4100, 1041, 256, 1088)])
def test_gemm_bf16_forward_exact_256_tile_form(M, N, K, ldc):
    """M >= 4096 and N % 256 == 0 take the 256 x 256 tile kernel (ragged last row tile, ldc > N, several tiles per
    workgroup); N = 320 (a badly filled 256-column tile) takes the wave-specialised LDS-DMA kernel, N = 1041 the 256 x 256
    one with a ragged last column tile; plus the row-segmented (expert) form with an empty and ragged segments, and +=."""
    g = torch.Generator().manual_seed(M + N + K)
    A, B = ints((M, K), g), ints((N, K), g)
    B = B + torch.arange(N)[:, None] % 3
    C0 = ints((M, ldc), g, -2, 3)
    c = dev(C0.to(BF))
    ops.gemm(dev(A.to(BF)), K, 1, dev(B.to(BF)), K, 1, c, ldc, M, N, K)
    ref = A.double() @ B.double().T
    assert torch.equal(c.cpu()[:, :N].float(), ref.to(BF).float())
    assert torch.equal(c.cpu()[:, N:].float(), C0[:, N:].to(BF).float()), "columns past N must stay untouched"
    c2 = dev(C0.to(BF))
    ops.gemm(dev(A.to(BF)), K, 1, dev(B.to(BF)), K, 1, c2, ldc, M, N, K, accumulate=True)
    assert torch.equal(c2.cpu()[:, :N].float(), (ref + C0[:, :N].double()).to(BF).float())
    if N == 512:
        E = 6
        seg = [0, 700, 700, 1725, 2000, 3300, M]
        W = ints((E * N, K), g)
        cg = torch.full((M, N), 7.0, dtype=BF, device=DEV)
        ops.gemm(dev(A.to(BF)), K, 1, dev(W.to(BF)), K, 1, cg, N, M, N, K, groups=E,
                 group_offsets=dev(torch.tensor(seg, dtype=torch.int32)), strideB=N * K)
        refg = torch.cat([A[seg[e]:seg[e + 1]].double() @ W[e * N:(e + 1) * N].double().T for e in range(E)])
        assert torch.equal(cg.cpu().float(), refg.to(BF).float())


def test_gemm_bf16_grouped_rows():
    g = torch.Generator().manual_seed(3)
    K, N, E = 256, 256, 6
    seg = [0, 130, 130, 400, 657, 900, 1000]                    # an empty group, ragged tiles
    M = seg[-1]
    A, W = ints((M, K), g), ints((E * N, K), g)
    a16, w16 = dev(A.to(BF)), dev(W.to(BF))
    c = torch.full((M, N), 7.0, dtype=BF, device=DEV)
    offs = dev(torch.tensor(seg, dtype=torch.int32))
    ops.gemm(a16, K, 1, w16, K, 1, c, N, M, N, K, groups=E, group_offsets=offs, strideB=N * K)
    ref = torch.cat([A[seg[e]:seg[e + 1]].double() @ W[e * N:(e + 1) * N].double().T for e in range(E)])
    assert torch.equal(c.cpu().float(), ref.to(BF).float())


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("use_map", [False, True])
def test_gemm_bf16_residual_epilogue_matches_fp32_kernel(p, use_map):
    """fp32 residual stream out = resid + dropout(acc): identical to gamer_gemm_f32's fused epilogue on the same
    (exact) data, same seed - so the backward kernels regenerate the mask of either path."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 384, 256, 128
    A, W, R = ints((M, K), g), ints((N, K), g), torch.randn(M, N, generator=g)
    perm = dev(torch.randperm(M, generator=g).int()) if use_map else None
    out16 = torch.zeros(M, N, device=DEV)
    out32 = torch.zeros(M, N, device=DEV)
    ops.gemm(dev(A.to(BF)), K, 1, dev(W.to(BF)), K, 1, out16, N, M, N, K, resid=dev(R), row_map=perm, p_drop=p, seed=4242)
    ops.gemm(dev(A), K, 1, dev(W), K, 1, out32, N, M, N, K, resid=dev(R), row_map=perm, p_drop=p, seed=4242)
    assert torch.equal(out16, out32)
    if p == 0.0 and not use_map:
        assert torch.allclose(out16.cpu().double(), R.double() + A.double() @ W.double().T, atol=1e-5)


@pytest.mark.parametrize("M,N,rows,lda,ldb", [(128, 128, 64, 128, 128), (768, 256, 1000, 768, 256), (1041, 256, 517, 1088, 256),
                                              (512, 320, 300, 512, 320), (256, 512, 4100, 256, 512)])
def test_gemm_bf16_wgrad_exact(M, N, rows, lda, ldb):
    g = torch.Generator().manual_seed(M + rows)
    dY = ints((rows, lda), g, -2, 3)
    X = ints((rows, ldb), g, -2, 3) + (torch.arange(ldb)[None, :] % 2)
    dW = torch.full((M, N), 0.5, device=DEV)
    ops.linear_wgrad(dev(dY.to(BF)), lda, dev(X.to(BF)), ldb, dW, N, rows, M, N, kchunk=128)
    ref = dY[:, :M].double().T @ X[:, :N].double() + 0.5
    assert torch.equal(dW.cpu().double(), ref)


def test_gemm_bf16_wgrad_grouped():
    g = torch.Generator().manual_seed(9)
    E, M, N = 6, 256, 320
    seg = [0, 70, 70, 500, 1024, 1100, 1500]
    rows = seg[-1]
    dY, X = ints((rows, M), g, -2, 3), ints((rows, N), g, -2, 3)
    dW = torch.zeros(E * M, N, device=DEV)
    offs = dev(torch.tensor(seg, dtype=torch.int32))
    ops.linear_wgrad(dev(dY.to(BF)), M, dev(X.to(BF)), N, dW, N, rows, M, N, groups=E, group_offsets=offs, strideC=M * N,
                     kchunk=256)
    ref = torch.cat([dY[seg[e]:seg[e + 1]].double().T @ X[seg[e]:seg[e + 1]].double() for e in range(E)])
    assert torch.equal(dW.cpu().double(), ref)


def test_gemm_bf16_rowdot_epilogue():
    g = torch.Generator().manual_seed(5)
    B_, S, nq, K = 2, 128, 2, 256
    T, NQ = B_ * S, nq * 64
    dY, WT = ints((T, K), g, -2, 3), ints((NQ, K), g, -2, 3)
    O = ints((T, NQ), g, -2, 3)
    dao = torch.zeros(T, NQ, dtype=BF, device=DEV)
    delta = torch.zeros(B_, nq, S, device=DEV)
    ops.linear_dgrad_t(dev(dY.to(BF)), K, dev(WT.to(BF)), K, dao, NQ, T, K, NQ, rowdot=(dev(O.to(BF)), delta, S))
    ref = (dY.double() @ WT.double().T).to(BF)
    assert torch.equal(dao.cpu().float(), ref.float())
    d_ref = (ref.double() * O.double()).view(B_, S, nq, 64).sum(-1).permute(0, 2, 1)
    assert torch.allclose(delta.cpu().double(), d_ref, rtol=1e-6, atol=1e-3)


def test_cast_params_and_transposes():
    from gamer_amd.config import synthetic_config
    from gamer_amd.engine import Bf16Shadow, ParamLayout
    cfg = synthetic_config(num_hidden_layers=2, behavior_injection_decoder=[0], cross_attention_decoder=[1], codebook=8)
    layout = ParamLayout(cfg)
    flat = torch.randn(layout.numel, device=DEV)
    sh = Bf16Shadow(cfg, layout, flat)
    sh.refresh()
    views = layout.views(flat)
    for key in ("model.embed_tokens.weight", "model.layers.1.cross_attn.gating.weight", "model.layers.0.self_attn.o_proj.weight",
                "model.layers.1.mlp.experts.expert_5.down_proj.weight"):
        w = views[key]
        assert torch.equal(sh.params16[key], w.to(BF))
        t = sh.t(key)
        assert torch.equal(t[:, :w.shape[0]], w.to(BF).T)
        assert bool((t[:, w.shape[0]:] == 0).all())
    assert torch.equal(sh.params16["model.layers.0.mlp.experts.expert_3.up_proj.weight"],
                       views["model.layers.0.mlp.experts.expert_3.up_proj.weight"].to(BF))
    qkv = torch.cat([views[f"model.layers.1.cross_attn.{n}_proj.weight"] for n in "qkv"])
    assert torch.equal(sh.t("model.layers.1.cross_attn.qkv"), qkv.to(BF).T)
    # gate_e | up_e of an expert are one [2 I, din] matrix (the fused projection): its transpose is [din, 2 I]
    gu3 = torch.cat([views[f"model.layers.0.mlp.experts.expert_3.{n}_proj.weight"] for n in ("gate", "up")])
    assert torch.equal(sh.t("model.layers.0.mlp.experts.expert_3.gu"), gu3.to(BF).T)
    # expert e sits e * rows * cols behind expert 0 in the transposed buffer (grouped dgrad stride)
    t0 = sh.t("model.layers.0.mlp.experts.expert_0.gu")
    t2 = sh.t("model.layers.0.mlp.experts.expert_2.gu")
    assert t2.data_ptr() - t0.data_ptr() == 2 * t0.numel() * 2


# ----------------------------------------------------------------------------------------------------------------
def _bfr(t):
    """values representable in bf16, as fp32"""
    return t.to(BF).float()


def test_elementwise_twins_match_fp32_kernels_rounded_once():
    g = torch.Generator().manual_seed(0)
    T, H, I = 300, 256, 512
    x, w = torch.randn(T, H, generator=g), torch.rand(H, generator=g) + 0.5
    # rmsnorm fwd
    y32 = torch.empty(T, H, device=DEV)
    y16 = torch.empty(T, H, dtype=BF, device=DEV)
    ops.rmsnorm_fwd(dev(x), dev(w), 1e-6, y32)
    ops.rmsnorm_fwd(dev(x), dev(w), 1e-6, y16)
    assert torch.equal(y16, y32.to(BF))
    # rmsnorm bwd (dy bf16-representable): dx identical, mask_out rounded once
    dy = _bfr(torch.randn(T, H, generator=g))
    for p in (0.0, 0.2):
        dx32, dx16 = torch.zeros(T, H, device=DEV), torch.zeros(T, H, device=DEV)
        pa32, pa16 = torch.empty(64, H, device=DEV), torch.empty(64, H, device=DEV)
        m32, m16 = torch.empty(T, H, device=DEV), torch.empty(T, H, dtype=BF, device=DEV)
        ops.rmsnorm_bwd(dev(x), dev(w), dev(dy), H, 1e-6, dx32, pa32, False, mask_out=m32, p=p, seed=5)
        ops.rmsnorm_bwd(dev(x), dev(w), dev(dy.to(BF)), H, 1e-6, dx16, pa16, False, mask_out=m16, p=p, seed=5)
        assert torch.equal(dx16, dx32) and torch.equal(pa16, pa32) and torch.equal(m16, m32.to(BF))
    # swiglu fwd / bwd
    gg, uu, dh = (_bfr(torch.randn(T, I, generator=g)) for _ in range(3))
    h32, h16 = torch.empty(T, I, device=DEV), torch.empty(T, I, dtype=BF, device=DEV)
    ops.swiglu_fwd(dev(gg), dev(uu), T * I, 0.2, 9, h32)
    ops.swiglu_fwd(dev(gg.to(BF)), dev(uu.to(BF)), T * I, 0.2, 9, h16)
    assert torch.equal(h16, h32.to(BF))
    g32, u32, g16, u16 = dev(gg), dev(uu), dev(gg.to(BF)), dev(uu.to(BF))
    ops.swiglu_bwd(g32, u32, dev(dh), T * I, 0.2, 9)
    ops.swiglu_bwd(g16, u16, dev(dh.to(BF)), T * I, 0.2, 9)
    assert torch.equal(g16, g32.to(BF)) and torch.equal(u16, u32.to(BF))
    # output gate + residual
    a, gt, res, dout = _bfr(torch.randn(T, H, generator=g)), _bfr(torch.randn(T, H, generator=g)), torch.randn(T, H, generator=g), \
        torch.randn(T, H, generator=g)
    o32, o16 = torch.empty(T, H, device=DEV), torch.empty(T, H, device=DEV)
    ops.silu_gate_fwd(dev(a), dev(gt), o32, resid=dev(res), p=0.2, seed=3)
    ops.silu_gate_fwd(dev(a.to(BF)), dev(gt.to(BF)), o16, resid=dev(res), p=0.2, seed=3)
    assert torch.equal(o16, o32)
    da32, dg32 = torch.empty(T, H, device=DEV), torch.empty(T, H, device=DEV)
    da16, dg16 = torch.empty(T, H, dtype=BF, device=DEV), torch.empty(T, H, dtype=BF, device=DEV)
    ops.silu_gate_bwd(dev(a), dev(gt), dev(dout), da32, dg32, p=0.2, seed=3)
    ops.silu_gate_bwd(dev(a.to(BF)), dev(gt.to(BF)), dev(dout), da16, dg16, p=0.2, seed=3)
    assert torch.equal(da16, da32.to(BF)) and torch.equal(dg16, dg32.to(BF))
    # behaviour-table rows
    tbl, idx = torch.randn(4, 64, generator=g), torch.randint(0, 4, (T,), generator=g).int()
    y32, y16 = torch.zeros(T, 320, device=DEV), torch.zeros(T, 320, dtype=BF, device=DEV)
    ops.rowtable_fwd(dev(tbl), dev(idx), y32, 320, 256)
    ops.rowtable_fwd(dev(tbl), dev(idx), y16, 320, 256)
    assert torch.equal(y16, y32.to(BF))
    dyt = _bfr(torch.randn(T, 320, generator=g))
    d32, d16 = torch.zeros(4, 64, device=DEV), torch.zeros(4, 64, device=DEV)
    ops.rowtable_bwd(dev(dyt), 320, 256, dev(idx), d32)
    ops.rowtable_bwd(dev(dyt.to(BF)), 320, 256, dev(idx), d16)
    assert torch.allclose(d16, d32, rtol=1e-5, atol=1e-5)


def test_ce_bf16_matches_reference_arithmetic():
    g = torch.Generator().manual_seed(2)
    B_, S, V, ldl, temp = 3, 10, 1041, 1088, 0.7
    z = (torch.randn(B_ * S, ldl, generator=g) * 2).to(BF)
    z[:, V:] = 0
    labels = torch.randint(0, V, (B_, S), generator=g)
    labels[0, 3] = -100
    lg = dev(z.clone())
    lse, rl = torch.empty(B_ * S, device=DEV), torch.empty(B_ * S, device=DEV)
    ls, cnt = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    ops.ce_fwd(lg, ldl, dev(labels), V, temp, -100, lse, rl, ls, cnt)
    zs = (z[:, :V].float() / temp).to(BF)                                   # in-place bf16 division, as upstream
    assert torch.equal(lg.cpu()[:, :V], zs) and bool((lg.cpu()[:, V:] == 0).all())
    shift = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1)
    ref = torch.nn.functional.cross_entropy(zs.float(), shift, ignore_index=-100, reduction="sum")
    assert abs(float(ls) - float(ref)) < 1e-4 * float(ref) and float(cnt) == float((shift != -100).sum())
    ops.ce_bwd(lg, ldl, dev(labels), V, temp, -100, lse, cnt, 0.0, 1.0)
    zz = zs.float().requires_grad_(True)
    torch.nn.functional.cross_entropy(zz, shift, ignore_index=-100, reduction="mean").backward()
    got = lg.cpu()[:, :V].float()
    want = (zz.grad / temp)
    assert float((got - want).abs().max()) <= 2 ** -8 * float(want.abs().max()) + 1e-9
    assert bool((lg.cpu()[:, V:] == 0).all())


@pytest.mark.parametrize("B_,S", [(2, 128), (16, 40), (128, 505), (256, 505)])
@pytest.mark.parametrize("cross", [False, True])
def test_gemm_bf16_qkv_epilogue_equals_gemm_then_qknorm_rope(cross, B_, S):
    """The bf16 q|k|v projection with the per-head RMSNorm + RoPE epilogue (gamer_gemm_bf16_desc.qk_*; both tile forms: the
    last shape takes the 256 x 256 kernel) against the two-kernel form: bit-identical raw q|k|v (v + bias_v in the cross
    attention), rotated q / k within one bf16 ulp (the 64-wide sum of squares is grouped differently).  256 x 505 tokens take
    the 256 x 256 tile form, 128 x 505 the 128 x 128 one."""
    nq, nkv, H, nb1 = 6, 3, 256, 4
    T, QKV = B_ * S, (nq + 2 * nkv) * 64
    assert T % 128 == 0
    g = torch.Generator().manual_seed(23)
    x, W = _bfr(torch.randn(T, H, generator=g)).to(BF), (_bfr(torch.randn(QKV, H, generator=g) * 0.1)).to(BF)
    wq, wk = 1 + 0.1 * torch.randn(64, generator=g), 1 + 0.1 * torch.randn(64, generator=g)
    cos, sin = orc.rope_tables(S, 64, 1e6)
    pos_ids = dev(torch.randint(0, S, (T,), generator=g).int()) if (cross and S == 40) else None
    bias = dict(bias_q=dev(torch.randn(nb1, nq * 64, generator=g)), bias_k=dev(torch.randn(nb1, nkv * 64, generator=g)),
                bias_v=dev(torch.randn(nb1, nkv * 64, generator=g)),
                act_idx=dev(torch.randint(0, nb1, (T,), generator=g).int())) if cross else {}
    ref = torch.empty(T, QKV, dtype=BF, device=DEV)
    q_ref, k_ref = torch.empty(T, nq * 64, dtype=BF, device=DEV), torch.empty(T, nkv * 64, dtype=BF, device=DEV)
    ops.linear_fwd(dev(x), H, dev(W), H, ref, QKV, T, QKV, H)
    ops.qknorm_rope_fwd(ref, S, nq, nkv, dev(wq), dev(wk), 1e-6, dev(cos), dev(sin), q_ref, k_ref, pos_ids=pos_ids, **bias)
    out = torch.full((T, QKV), float("nan"), dtype=BF, device=DEV)
    q_rot, k_rot = torch.full_like(q_ref, float("nan")), torch.full_like(k_ref, float("nan"))
    assert ops.qkv_fused_ok(dev(x), T, QKV)
    ops.gemm(dev(x), H, 1, dev(W), H, 1, out, QKV, T, QKV, H,
             qknorm=dict(wq=dev(wq), wk=dev(wk), eps=1e-6, cos=dev(cos), sin=dev(sin), q_rot=q_rot, k_rot=k_rot, pos_ids=pos_ids,
                         S=S, nq=nq, nkv=nkv, **bias))
    assert torch.equal(out, ref)                      # (the two-kernel form's v + bias_v is written back in place as well)

    def ulps(a, b, heads):
        """largest difference in units of the bf16 spacing at the magnitude of the rotated PAIR (d, d + 32): RoPE keeps the
        pair's norm, and a component can cancel to far below it"""
        a, b = a.float().view(T, heads, 2, 32), b.float().view(T, heads, 2, 32)
        mag = b.pow(2).sum(2, keepdim=True).sqrt().clamp_min(2.0 ** -6)
        spacing = torch.exp2(torch.floor(torch.log2(mag)) - 7)
        return float(((a - b).abs() / spacing).max())
    # one rounding may flip; in the self attention the normalised value is rounded before the weight as well: two
    assert ulps(q_rot, q_ref, nq) <= (1 if cross else 2) and ulps(k_rot, k_ref, nkv) <= (1 if cross else 2)
    assert float((q_rot.float() - q_ref.float()).abs().max()) < 2e-2 * float(q_ref.float().abs().max())
    mism = float((q_rot != q_ref).float().mean())
    assert mism < 0.02, mism                           # a different summation grouping flips a rounding now and then, no more


@pytest.mark.parametrize("cross", [False, True])
def test_qknorm_rope_bf16(cross):
    """Against a torch restatement with the casts autocast makes (see the kernel's header comment), fwd and bwd."""
    g = torch.Generator().manual_seed(4)
    B_, S, nq, nkv, NB1 = 2, 15, 2, 1, 4
    T, ld = B_ * S, (nq + 2 * nkv) * 64
    qkv = _bfr(torch.randn(T, ld, generator=g))
    wq, wk = torch.rand(64, generator=g) + 0.5, torch.rand(64, generator=g) + 0.5
    cos, sin = orc.rope_tables(S, 64, 1e6)
    act = torch.randint(0, NB1, (T,), generator=g).int()
    bq, bk, bv = (torch.randn(NB1, n * 64, generator=g) * 0.3 for n in (nq, nkv, nkv))
    dq_r, dk_r = _bfr(torch.randn(T, nq * 64, generator=g)), _bfr(torch.randn(T, nkv * 64, generator=g))

    def ref(qkv_, wq_, wk_, bq_, bk_, bv_):
        q = qkv_[:, :nq * 64].view(B_, S, nq, 64)
        k = qkv_[:, nq * 64:(nq + nkv) * 64].view(B_, S, nkv, 64)
        v = qkv_[:, (nq + nkv) * 64:].view(B_, S, nkv, 64)
        if cross:
            a = act.long().view(B_, S)
            q, k, v = q + bq_[a].view(B_, S, nq, 64), k + bk_[a].view(B_, S, nkv, 64), v + bv_[a].view(B_, S, nkv, 64)

        def norm(x, w):
            xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
            if not cross:
                xn = xn + (xn.to(BF).to(xn.dtype) - xn).detach()          # rounded value, identity gradient
            return w * xn
        return orc.apply_rope(norm(q, wq_), cos.double(), sin.double()), orc.apply_rope(norm(k, wk_), cos.double(), sin.double()), v

    leaves = [t.double().requires_grad_(True) for t in (qkv, wq, wk, bq, bk, bv)]
    q_ref, k_ref, v_ref = ref(*leaves)
    (q_ref.reshape(T, -1) * dq_r.double()).sum().backward(retain_graph=True)
    (k_ref.reshape(T, -1) * dk_r.double()).sum().backward()
    d_qkv = dev(qkv.to(BF))
    q_rot, k_rot = torch.empty(T, nq * 64, dtype=BF, device=DEV), torch.empty(T, nkv * 64, dtype=BF, device=DEV)
    kw = dict(bias_q=dev(bq), bias_k=dev(bk), bias_v=dev(bv), act_idx=dev(act)) if cross else {}
    ops.qknorm_rope_fwd(d_qkv, S, nq, nkv, dev(wq), dev(wk), 1e-6, dev(cos), dev(sin), q_rot, k_rot, **kw)
    ulp = 2 ** -8
    assert _rel(q_rot, q_ref.reshape(T, -1)) < 1.5 * ulp and _rel(k_rot, k_ref.reshape(T, -1)) < 1.5 * ulp
    if cross:
        assert _rel(d_qkv[:, (nq + nkv) * 64:], v_ref.reshape(T, -1)) < 1.5 * ulp
        assert torch.equal(d_qkv[:, :(nq + nkv) * 64].cpu(), qkv[:, :(nq + nkv) * 64].to(BF)), "q/k stay un-biased in bf16"
    dqkv = torch.zeros(T, ld, dtype=BF, device=DEV)
    dwq, dwk = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    dbq, dbk, dbv = (torch.zeros(NB1, n * 64, device=DEV) for n in (nq, nkv, nkv))
    bkw = dict(bias_q=dev(bq), bias_k=dev(bk), act_idx=dev(act), nb1=NB1, dbias_q=dbq, dbias_k=dbk, dbias_v=dbv) if cross else {}
    ops.qknorm_rope_bwd(dev(qkv.to(BF)), dev(dq_r.to(BF)), dev(dk_r.to(BF)), S, nq, nkv, dev(wq), dev(wk), 1e-6, dev(cos),
                        dev(sin), dqkv, dwq, dwk, **bkw)
    gq = leaves[0].grad[:, :(nq + nkv) * 64]
    assert _rel(dqkv[:, :(nq + nkv) * 64], gq) < 2 * ulp
    assert _rel(dwq, leaves[1].grad) < 1e-3 and _rel(dwk, leaves[2].grad) < 1e-3
    if cross:
        assert _rel(dbq, leaves[3].grad) < 1e-3 and _rel(dbk, leaves[4].grad) < 1e-3


# ----------------------------------------------------------------------------------------------------------------
def _attn_ref_zero(q, k, v, ok, nq, nkv, scale, mult=None):
    """Dense fp64 reference of the bf16 run: rows without an allowed key give 0 and pass no gradient."""
    rep = nq // nkv
    kq, vq = k.repeat_interleave(rep, 2), v.repeat_interleave(rep, 2)
    s = torch.einsum("bind,bjnd->bnij", q, kq) * scale
    empty = ~ok.any(-1)
    s = s.masked_fill(~ok[:, None], float("-inf"))
    s = torch.where(empty[:, None, :, None], torch.zeros_like(s), s)
    p = torch.softmax(s, -1) * (~empty)[:, None, :, None]
    lse = torch.logsumexp(s, -1)
    pd = p if mult is None else p * mult
    return torch.einsum("bnij,bjnd->bind", pd, vq), lse, empty


def _run_attn16(batch, cross, B, S, nq, nkv, q, k, v, d_o=None, p_drop=0.0, seed=1234, ordered=False):
    T = B * S
    router = ops.alloc_router_outputs(B, S, DEV)
    lut = torch.full((64,), -1, dtype=torch.int32)
    ops.router_fwd(dev(batch["input_ids"]), dev(batch["attention_mask"]), dev(batch["actions"]), dev(lut), 5, 4, 8, router)
    kl = router["kl_cross"] if cross else router["kl_self"]
    ql = router["ql_cross"] if cross else None
    ldv = (nq + 2 * nkv) * 64
    qkv = torch.zeros(T, ldv, dtype=BF, device=DEV)
    qkv[:, (nq + nkv) * 64:] = dev(v.reshape(T, -1).to(BF))
    vview = qkv[:, (nq + nkv) * 64:]
    o = torch.full((T, nq * 64), float("nan"), dtype=BF, device=DEV)
    lse = torch.empty(B, nq, S, device=DEV)
    dq_, dk_ = dev(q.reshape(T, -1).to(BF)), dev(k.reshape(T, -1).to(BF))
    order = None
    if ordered:             # rows with an allowed key first (gamer_attn_row_order); results must not depend on it
        rempty = router["empty_cross"] if cross else router["empty_self"]
        n_t = (S + 31) // 32
        perm, kind, maxpos = (torch.empty(B, S, dtype=torch.int32, device=DEV), torch.empty(B, n_t, dtype=torch.int32, device=DEV),
                              torch.empty(B, n_t, dtype=torch.int32, device=DEV))
        ops.attn_row_order(rempty, perm, kind, maxpos)
        order = (perm, maxpos, rempty)
    ops.attn_fwd_bf16(dq_, nq * 64, dk_, nkv * 64, vview, ldv, kl, ql, B, S, nq, nkv, 0.125, p_drop, seed, o, lse, order=order)
    res = dict(o=o, lse=lse)
    if d_o is not None:
        delta = torch.empty(B, nq, S, device=DEV)
        dq = torch.full((T, nq * 64), float("nan"), dtype=BF, device=DEV)
        dk = torch.full((T, nkv * 64), float("nan"), dtype=BF, device=DEV)
        dqkv = torch.zeros(T, ldv, dtype=BF, device=DEV)
        dvv = dqkv[:, (nq + nkv) * 64:]
        ops.attn_bwd_bf16(dq_, nq * 64, dk_, nkv * 64, vview, ldv, o, dev(d_o.reshape(T, -1).to(BF)), lse, kl, ql, B, S, nq,
                          nkv, 0.125, p_drop, seed, delta, dq, nq * 64, dk, nkv * 64, dvv, ldv, order=order)
        res.update(dq=dq, dk=dk, dv=dvv, delta=delta)
    return res


@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("n_items,B,nq,nkv", [(7, 3, 2, 1), (14, 2, 2, 1), (20, 2, 2, 1), (101, 2, 2, 1), (101, 11, 6, 3),
                                              (33, 40, 6, 3), (26, 3, 3, 3)])
def test_attention_bf16_fwd_bwd(cross, n_items, B, nq, nkv):
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=7 + n_items, pad_rows={0: max(1, n_items // 3)})
    S = batch["input_ids"].shape[1]
    g = torch.Generator().manual_seed(n_items)
    q, k, v, d_o = (_bfr(torch.randn(B, S, n, 64, generator=g)) for n in (nq, nkv, nkv, nq))
    self_ok, cross_ok = orc.mask_predicates(batch["attention_mask"], batch["actions"])
    ok = cross_ok if cross else self_ok
    leaves = [t.double().requires_grad_(True) for t in (q, k, v)]
    o_ref, lse_ref, empty = _attn_ref_zero(*leaves, ok, nq, nkv, 0.125)
    (o_ref * d_o.double()).sum().backward()
    res = _run_attn16(batch, cross, B, S, nq, nkv, q, k, v, d_o)
    T = B * S
    if cross:
        assert int((empty & batch["attention_mask"].bool()).sum()) > 0, "fixture must contain empty rows"
        erows = empty.reshape(T)
        assert bool((res["o"].cpu()[erows] == 0).all()) and bool((res["dq"].cpu()[erows] == 0).all())
    e_o = _rel(res["o"], o_ref.reshape(T, -1))
    ne = ~empty
    e_l = float((res["lse"].cpu().permute(0, 2, 1)[ne].double() - lse_ref.permute(0, 2, 1)[ne]).abs().max())
    assert e_l < 1e-3
    e = [e_o, _rel(res["dq"], leaves[0].grad.reshape(T, -1)), _rel(res["dk"], leaves[1].grad.reshape(T, -1)),
         _rel(res["dv"], leaves[2].grad.reshape(T, -1))]
    # probabilities and dS are rounded to bf16 (2^-9 relative) before the second products, outputs once more
    assert max(e) < 2e-2, e
    # the row order (cross attention: rows without an allowed key sorted behind the others and skipped) changes which
    # workgroup computes a row, never the row: forward outputs and dQ bit for bit; dK / dV sum the same query tiles in
    # another tile partition (bf16 rounding of the result only)
    for p_drop in (0.0, 0.2):
        base = res if p_drop == 0.0 else _run_attn16(batch, cross, B, S, nq, nkv, q, k, v, d_o, p_drop=p_drop)
        ordd = _run_attn16(batch, cross, B, S, nq, nkv, q, k, v, d_o, p_drop=p_drop, ordered=True)
        assert torch.equal(ordd["o"], base["o"]) and torch.equal(ordd["lse"], base["lse"]) and torch.equal(ordd["dq"], base["dq"])
        assert _rel(ordd["dk"], base["dk"].double()) < 1e-2 and _rel(ordd["dv"], base["dv"].double()) < 1e-2


@pytest.mark.parametrize("cross", [False, True])
def test_attention_bf16_dropout_mask_consistent_fwd_bwd(cross):
    """Recover the keep-mask with q = k = 0 and V = one-hot(j) (S <= 64), check it is the fp32 kernels' mask for the same
    seed, then check forward and backward with that exact mask against the dense reference."""
    B, n_items, nq, nkv, p = 2, 12, 2, 1, 0.2
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=21, pad_rows={1: 3})
    S = batch["input_ids"].shape[1]
    assert S <= 64
    self_ok, cross_ok = orc.mask_predicates(batch["attention_mask"], batch["actions"])
    ok = cross_ok if cross else self_ok
    zq, zk = torch.zeros(B, S, nq, 64), torch.zeros(B, S, nkv, 64)
    eye = torch.zeros(B, S, nkv, 64)
    eye[:, torch.arange(S), 0, torch.arange(S)] = 1.0
    r0 = _run_attn16(batch, cross, B, S, nq, nkv, zq, zk, eye, p_drop=p, seed=99)
    pd = r0["o"].cpu().float().view(B, S, nq, 64)[..., :S].permute(0, 2, 1, 3).double()     # [B,nq,S,S] = dropped p
    cnt = ok.sum(-1).clamp_min(1).double()
    punif = ok.double() / cnt[..., None]
    mult = torch.where(punif[:, None] > 0, pd / punif[:, None].clamp_min(1e-30), torch.zeros_like(pd))
    vals = mult[(punif[:, None] > 0).expand_as(mult)]
    assert bool(((vals - 1.25).abs() < 2e-2).logical_or(vals.abs() < 1e-7).all()), "mask values must be 0 or 1/(1-p)"
    assert abs(float((vals > 0).double().mean()) - 0.8) < 0.03
    mult = torch.where(mult > 0.5, torch.full_like(mult, 1.25), torch.zeros_like(mult))
    g = torch.Generator().manual_seed(5)
    q, k, v, d_o = (_bfr(torch.randn(B, S, n, 64, generator=g)) for n in (nq, nkv, nkv, nq))
    leaves = [t.double().requires_grad_(True) for t in (q, k, v)]
    o_ref, _, _ = _attn_ref_zero(*leaves, ok, nq, nkv, 0.125, mult=mult)
    (o_ref * d_o.double()).sum().backward()
    res = _run_attn16(batch, cross, B, S, nq, nkv, q, k, v, d_o, p_drop=p, seed=99)
    T = B * S
    e = [_rel(res["o"], o_ref.reshape(T, -1)), _rel(res["dq"], leaves[0].grad.reshape(T, -1)),
         _rel(res["dk"], leaves[1].grad.reshape(T, -1)), _rel(res["dv"], leaves[2].grad.reshape(T, -1))]
    assert max(e) < 2e-2, e


# ----------------------------------------------------------------------------------------------------------------
def _engine16(golden, name, dropout=False):
    from gamer_amd.config import Qwen3MultiConfig
    from gamer_amd.engine import Engine
    z, meta = golden(name)
    cfg = Qwen3MultiConfig.from_dict(meta["config"])
    if not dropout:
        cfg.dropout_rate = cfg.attention_dropout = 0.0
    ocfg = orc.OracleConfig.from_dict(meta["config"])
    sd = orc.init_state_dict(ocfg, seed=meta["weight_seed"])
    eng = Engine(cfg, device=DEV, temperature=meta["temperature"], dtype="bf16")
    eng.load_state_dict(sd)
    batch = {k: torch.from_numpy(z[k]) for k in ("input_ids", "attention_mask", "actions", "labels")}
    return z, meta, eng, batch, sd, ocfg


@pytest.mark.parametrize("name", ["small_bf16", "full_bf16"])
def test_model_bf16_against_reference_autocast_fixture(golden, name):
    """The shipped path in bf16 against the reference executed under torch.autocast(bfloat16) (its --bf16 run).
    Tolerances: tests/test_oracle.py::test_amp_forward_and_gradients measured the noise between two implementations of
    this arithmetic (4e-3 of the logits' abs-max, 5e-3 on gradient norms, 3e-2 on single entries); the fp32 semantics
    (uniform empty rows) are 7.6e-2 / 1.5e-1 away on these fixtures, so they cannot pass."""
    z, meta, eng, batch, sd, ocfg = _engine16(golden, name)
    key = "logits_raw" if "logits_raw" in z.files else "logits_raw_sample"
    _, logits = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], train=False)
    assert logits.dtype == BF
    got = logits.float().cpu()
    if key.endswith("sample"):
        got = got[:, ::37, ::53]
    ref = torch.from_numpy(z[key])
    amax = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 1e-2 * amax
    loss, _ = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"], train=True,
                          dropout=False)
    assert abs(float(loss) - float(z["loss_train_mode"])) < 1e-3
    eng.zero_grad()
    eng.backward(1.0)
    torch.cuda.synchronize()
    eng.check_inputs()
    gkeys = [str(k) for k in z["grad_keys"]]
    norms = np.array([float(eng.grads[k].double().norm()) for k in gkeys])
    np.testing.assert_allclose(norms, z["grad_norms"], rtol=3e-2, atol=1e-9)
    gn = float(np.sqrt((norms ** 2).sum()))
    assert abs(gn - float(z["global_grad_norm"])) < 5e-3 * float(z["global_grad_norm"])
    for k in z.files:
        if k.startswith("grad::") or k.startswith("gradsample::"):
            gt = eng.grads[k.split("::")[1]].cpu()
            got = gt.numpy() if k.startswith("grad::") else gt[::max(1, gt.shape[0] // 8), ::max(1, gt.shape[1] // 8)].numpy()
            scale = max(np.abs(z[k]).max(), 1e-12)
            assert np.abs(got - z[k]).max() <= 8e-2 * scale, k


def test_model_bf16_against_amp_oracle_and_train_step(golden):
    """bf16 engine vs the CPU oracle's AMP restatement on a ragged batch, then a few optimizer steps with dropout on:
    the loss must fall like the fp32 engine's (same seeds, same data)."""
    from gamer_amd.config import synthetic_config
    from gamer_amd.engine import Engine
    cfg = synthetic_config(codebook=8, num_hidden_layers=4, behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3])
    ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
    sd = orc.init_state_dict(ocfg, seed=3)
    batch = synthetic.make_batch(6, 9, 8, 3, ragged=True, seed=8)
    c0 = synthetic_config(codebook=8, num_hidden_layers=4, behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3])
    c0.dropout_rate = c0.attention_dropout = 0.0
    eng = Engine(c0, device=DEV, temperature=0.7, dtype="bf16")
    eng.load_state_dict(sd)
    loss, logits = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"], train=True)
    lg = logits.float().cpu()
    eng.zero_grad()
    eng.backward(1.0)
    ref_loss, ref_grads, ref_out = orc.loss_and_grads(sd, ocfg, batch, temperature=0.7, amp=True)
    rl = ref_out["logits"].detach().float()
    assert float((lg - rl).abs().max()) < 1.5e-2 * float(rl.abs().max())
    assert abs(float(loss) - float(ref_loss)) < 1e-3
    for k_, g_ in ref_grads.items():
        n_ref = float(g_.double().norm())
        assert abs(float(eng.grads[k_].double().norm()) - n_ref) <= 3e-2 * n_ref + 1e-7, k_
    losses = {}
    for dt in ("f32", "bf16"):
        e2 = Engine(cfg, device=DEV, temperature=0.7, dtype=dt)
        e2.load_state_dict(sd)
        ls = []
        for step in range(12):
            b = synthetic.make_batch(16, 9, 8, 3, seed=100 + step % 3)
            ls.append(float(e2.train_step(b, 2e-3)))
        e2.check_inputs()
        losses[dt] = ls
    assert losses["bf16"][-1] < losses["bf16"][0] - 0.2
    assert all(math.isfinite(x) for x in losses["bf16"])
    assert abs(losses["bf16"][-1] - losses["f32"][-1]) < 0.15, losses
