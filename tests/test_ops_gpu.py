"""Per-kernel parity: every C-ABI entry point against a CPU reference on the same seeded inputs.

Integer work (router, expert lists) is bit-exact; floating point is compared to fp64/fp32 torch
CPU math with the tolerance written at each assert.  All calls go through the C ABI
(gamer_amd.ops -> ctypes -> libgamer_hip.so).
"""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import ops, synthetic  # noqa: E402
from oracle import qwen3multi_oracle as orc  # noqa: E402

DEV = "cuda"
REPORT = {}


def _rel(got: torch.Tensor, ref: torch.Tensor) -> float:
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _record(name, value):
    REPORT[name] = value
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "ops_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    torch.manual_seed(0)


def dev(t):
    return t.to(DEV).contiguous()


# ----------------------------------------------------------------------------------------------
def test_router_and_expert_lists_bit_exact():
    cb, NB = 8, 3
    cfg = orc.OracleConfig(vocab_size=synthetic.vocab_size(cb, NB), behavior_maps=synthetic.behavior_maps(cb, NB))
    batch = synthetic.make_batch(7, 13, cb, NB, ragged=True, seed=3)
    ids = batch["input_ids"].clone()
    ids[2, 10:15] = cfg.eos_token_id              # an eos-only item in the middle
    B, S = ids.shape
    lut = torch.full((cfg.vocab_size,), -1, dtype=torch.int32)
    for tok, b in cfg.behavior_maps.items():
        lut[tok] = b
    out = ops.alloc_router_outputs(B, S, DEV)
    ops.router_fwd(dev(ids), dev(batch["attention_mask"]), dev(batch["actions"]), dev(lut), 5, cfg.pad_token_id,
                   cfg.eos_token_id, out)
    pos, beh, act = orc.router(ids, cfg)
    assert torch.equal(out["expert"].cpu().long(), pos)
    assert torch.equal(out["beh_idx"].cpu().long(), beh)
    assert torch.equal(out["act_idx"].cpu().long(), act)
    self_ok, cross_ok = orc.mask_predicates(batch["attention_mask"], batch["actions"])
    assert torch.equal(out["empty_self"].cpu().bool(), ~self_ok.any(-1))
    assert torch.equal(out["empty_cross"].cpu().bool(), ~cross_ok.any(-1))
    # predicate reconstruction from the level arrays
    kl, ql = out["kl_cross"].cpu().long(), out["ql_cross"].cpu().long()
    i = torch.arange(S).view(1, S, 1)
    j = torch.arange(S).view(1, 1, S)
    assert torch.equal((j <= i) & (kl[:, None, :] < ql[:, :, None]), cross_ok)
    assert torch.equal((j <= i) & (out["kl_self"].cpu().long()[:, None, :] < 1), self_ok)
    n_tiles = (S + 31) // 32
    es = torch.nn.functional.pad((~cross_ok.any(-1)).int(), (0, n_tiles * 32 - S)).view(B, n_tiles, 32).amax(-1)
    assert torch.equal(out["tile_empty_cross"].cpu(), es.int())
    assert int(out["bad_token"].item()) == 0
    # expert lists
    E = 6
    perm = torch.empty(B * S, dtype=torch.int32, device=DEV)
    slot = torch.empty(B * S, dtype=torch.int32, device=DEV)
    offsets = torch.empty(E + 1, dtype=torch.int32, device=DEV)
    work = torch.empty((B + 1) * E, dtype=torch.int32, device=DEV)
    ops.expert_lists(out["expert"], E, perm, slot, offsets, work)
    flat = pos.reshape(-1)
    ref_perm = torch.sort(flat, stable=True).indices
    assert torch.equal(perm.cpu().long(), ref_perm)
    ref_slot = torch.empty_like(ref_perm)
    ref_slot[ref_perm] = torch.arange(B * S)
    assert torch.equal(slot.cpu().long(), ref_slot)
    counts = torch.bincount(flat, minlength=E)
    assert torch.equal(offsets.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]))


def test_router_flags_unknown_behavior_token():
    cb, NB = 8, 3
    cfg = orc.OracleConfig(vocab_size=synthetic.vocab_size(cb, NB), behavior_maps=synthetic.behavior_maps(cb, NB))
    batch = synthetic.make_batch(2, 4, cb, NB, seed=5)
    ids = batch["input_ids"].clone()
    ids[0, 5] = 20                                  # a semantic token where a behaviour token must be
    lut = torch.full((cfg.vocab_size,), -1, dtype=torch.int32)
    for tok, b in cfg.behavior_maps.items():
        lut[tok] = b
    out = ops.alloc_router_outputs(2, ids.shape[1], DEV)
    ops.router_fwd(dev(ids), None, dev(batch["actions"]), dev(lut), 5, 4, 8, out)
    assert int(out["bad_token"].item()) == 1


def test_embedding_fwd_bwd():
    V, H, T = 49, 128, 300
    W = torch.randn(V, H)
    ids = torch.randint(0, V, (T,))
    ids[:20] = 4
    x = torch.empty(T, H, device=DEV)
    ops.embedding_fwd(dev(ids), dev(W), x)
    assert torch.equal(x.cpu(), W[ids])
    dx = torch.randn(T, H)
    dW = torch.full((V, H), 0.5, device=DEV)
    ops.embedding_bwd(dev(ids), dev(dx), 4, dW)
    ref = torch.full((V, H), 0.5, dtype=torch.float64)
    m = ids != 4
    ref.index_put_((ids[m],), dx[m].double(), accumulate=True)
    assert _rel(dW, ref) < 1e-5


@pytest.mark.parametrize("V,H,T", [(49, 128, 300), (1041, 256, 5000), (7, 64, 1), (300, 1024, 2049)])
def test_embedding_bwd_ordered_is_exact_in_order_and_repeatable(V, H, T):
    """gamer_embedding_bwd_ordered: stable sort by id + sums in token order (pieces of 256 tokens).  Against an fp64 scatter-add,
    against the same sums formed in the kernel's order on the host (bit for bit on a small case), twice the same bits; padding
    and out-of-range ids are skipped; rows without tokens keep their value; a hot row (a fifth of the tokens) crosses pieces."""
    gen = torch.Generator().manual_seed(V + T)
    ids = torch.randint(0, V, (T,), generator=gen)
    ids[::5] = min(3, V - 1)                          # a hot row
    pad = 4 if V > 4 else 0
    if T > 10:
        ids[7], ids[8] = -1, V + 3                    # out of range: skipped
    dx = torch.randn(T, H, generator=gen)
    base = torch.randn(V, H, generator=gen)
    outs = []
    for _ in range(2):
        dW = dev(base.clone())
        ops.embedding_bwd_ordered(dev(ids), dev(dx), pad, dW)
        outs.append(dW.cpu())
    assert torch.equal(outs[0], outs[1])
    ok = (ids != pad) & (ids >= 0) & (ids < V)
    ref = base.double().clone()
    ref.index_put_((ids[ok],), dx[ok].double(), accumulate=True)
    assert _rel(outs[0], ref) < 1e-6
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[ids[ok]] = False
    assert torch.equal(outs[0][untouched], base[untouched])
    if T <= 300:
        # the kernel's order on the host: per row, token order; quarters of a piece summed left to right, then (q0 + q1) + (q2 + q3)
        want = base.clone()
        for v in range(V):
            tok = torch.nonzero(ok & (ids == v)).flatten()
            if len(tok) == 0:
                continue
            tot = None
            for p0 in range(0, len(tok), 256):
                piece = tok[p0:p0 + 256]
                qs = []
                for q in range(4):
                    acc = torch.zeros(H)
                    for t in piece[64 * q:64 * q + 64]:
                        acc = acc + dx[t]
                    qs.append(acc)
                ps = (qs[0] + qs[1]) + (qs[2] + qs[3])
                tot = ps if tot is None else tot + ps
            want[v] = want[v] + tot
        assert torch.equal(outs[0], want)


@pytest.mark.parametrize("H", [128, 256])
def test_rmsnorm_fwd_bwd(H):
    T, ldy = 333, H + 64
    x = torch.randn(T, H) * 2
    w = 1 + 0.1 * torch.randn(H)
    dst = torch.randperm(T).int()
    y = torch.zeros(T, ldy, device=DEV)
    ops.rmsnorm_fwd(dev(x), dev(w), 1e-6, y, ldy, dev(dst))
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = orc.rmsnorm(xr, wr, 1e-6)
    got = y.cpu()[dst.long(), :H]
    e = _rel(got, yr)
    _record(f"rmsnorm_fwd_H{H}", e)
    assert e < 2e-6
    dy = torch.randn(T, H)
    yr.backward(dy.double())
    dy_buf = torch.zeros(T, ldy)
    dy_buf[dst.long(), :H] = dy
    dx0 = torch.randn(T, H)
    dx = dev(dx0.clone())
    partial = torch.empty(64, H, device=DEV)
    ops.rmsnorm_bwd(dev(x), dev(w), dev(dy_buf), ldy, 1e-6, dx, partial, True, dev(dst))
    dw = torch.empty(H, device=DEV)
    ops.colsum_reduce(partial, dw)
    e1, e2 = _rel(dx.cpu() - dx0, xr.grad), _rel(dw, wr.grad)
    _record(f"rmsnorm_bwd_H{H}", [e1, e2])
    assert e1 < 1e-5 and e2 < 1e-5
    # fused branch gradient: mask_out == residual_dropout_bwd of the updated dx (same seed), plain and row-scattered
    for rows in (None, dev(torch.randperm(T).int())):
        dxa, dxb = dev(dx0.clone()), dev(dx0.clone())
        ma, mb = torch.zeros(T, H, device=DEV), torch.zeros(T, H, device=DEV)
        ops.rmsnorm_bwd(dev(x), dev(w), dev(dy_buf), ldy, 1e-6, dxa, partial, True, dev(dst), mask_out=ma,
                        mask_rows=rows, p=0.2, seed=123)
        ops.rmsnorm_bwd(dev(x), dev(w), dev(dy_buf), ldy, 1e-6, dxb, partial, True, dev(dst))
        ops.residual_dropout_bwd(dxb, 0.2, 123, mb, rows)
        assert torch.equal(dxa, dxb) and torch.equal(ma, mb)
        assert 0.15 < float((ma == 0).float().mean()) < 0.25


def test_rowtable_fwd_bwd():
    T, E, ld, col0, rows = 501, 64, 320, 256, 4
    table = torch.randn(rows, E)
    idx = torch.randint(0, rows, (T,)).int()
    dst = torch.randperm(T).int()
    y = torch.zeros(T, ld, device=DEV)
    ops.rowtable_fwd(dev(table), dev(idx), y, ld, col0, dev(dst))
    assert torch.equal(y.cpu()[dst.long(), col0:col0 + E], table[idx.long()])
    dy = torch.randn(T, ld)
    dt = torch.zeros(rows, E, device=DEV)
    ops.rowtable_bwd(dev(dy), ld, col0, dev(idx), dt, dev(dst))
    ref = torch.zeros(rows, E, dtype=torch.float64)
    ref.index_put_((idx.long(),), dy[dst.long(), col0:col0 + E].double(), accumulate=True)
    assert _rel(dt, ref) < 1e-5
    # the ordered form (a `partial` scratch: no float atomics): += semantics, the same bits on every launch, many and few workgroups
    for T2, scratch in ((T, 64 * 1024), (40000, 2048 * 256), (40000, 3 * rows * E)):
        idx2, dy2 = torch.randint(0, rows, (T2,)).int(), torch.randn(T2, ld)
        ref2 = torch.ones(rows, E, dtype=torch.float64)
        ref2.index_put_((idx2.long(),), dy2[:, col0:col0 + E].double(), accumulate=True)
        outs = []
        for rep in range(3):
            dt2 = torch.ones(rows, E, device=DEV)
            ops.rowtable_bwd(dev(dy2), ld, col0, dev(idx2), dt2, partial=torch.empty(scratch, device=DEV))
            outs.append(dt2)
        assert _rel(outs[0], ref2) < 1e-5 and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


# ----------------------------------------------------------------------------------------------
@pytest.fixture(params=["f32", "split3", "split6", "split9"])
def gemm_mode(request):
    """Every fp32 GEMM test runs on the fp32 MFMA, on both bf16-piece forms and on the two-way fp16 form (gamer_gemm_f32_split)
    with the SAME bars."""
    prev = ops.set_f32_matmul(request.param)
    yield request.param
    ops.set_f32_matmul(prev)


# (M >= 4096 with well-filled 256-column tiles: the bf16-piece forms take their 256 x 256 tile kernel there)
@pytest.mark.parametrize("M,N,K", [(300, 384, 256), (1000, 1041, 256), (257, 256, 1041), (128, 128, 32), (64, 40, 96),
                                   (4173, 256, 512), (4096, 1041, 256)])
def test_gemm_linear_fwd_and_dgrad(M, N, K, gemm_mode):
    ldx, ldw, ldy = (K + 3) // 4 * 4 + 8, (K + 3) // 4 * 4, (N + 3) // 4 * 4 + 4
    x = torch.zeros(M, ldx); x[:, :K] = torch.randn(M, K)
    w = torch.zeros(N, ldw); w[:, :K] = torch.randn(N, K)
    x[:, K:] = 7.0                                  # garbage beyond K must not leak in
    y = torch.full((M, ldy), -3.0, device=DEV)
    ops.linear_fwd(dev(x), ldx, dev(w), ldw, y, ldy, M, N, K)
    ref = x[:, :K].double() @ w[:, :K].double().T
    e = _rel(y.cpu()[:, :N], ref)
    _record(f"gemm_fwd_{M}x{N}x{K}_{gemm_mode}", e)
    assert e < 2e-6
    assert bool((y.cpu()[:, N:] == -3.0).all())
    # dgrad: dx[M,K] = dy[M,N] @ w[N,K]
    dy = torch.zeros(M, ldy); dy[:, :N] = torch.randn(M, N)
    dx = torch.zeros(M, ldx, device=DEV)
    ops.linear_dgrad(dev(dy), ldy, dev(w), ldw, dx, ldx, M, N, K)
    refd = dy[:, :N].double() @ w[:, :K].double()
    e = _rel(dx.cpu()[:, :K], refd)
    _record(f"gemm_dgrad_{M}x{N}x{K}_{gemm_mode}", e)
    assert e < 2e-6
    # accumulate
    ops.linear_dgrad(dev(dy), ldy, dev(w), ldw, dx, ldx, M, N, K, accumulate=True)
    assert _rel(dx.cpu()[:, :K], 2 * refd) < 2e-6


@pytest.mark.parametrize("B", [6, 64])
def test_gemm_rowdot_epilogue(gemm_mode, B):
    """dgrad layout with the row-dot epilogue: C as without it, and out[b][head][i] = C[m, head] . other[m, head]
    (flash attention's delta from the o_proj dgrad)."""
    S, heads, K = 64, 4, 96
    M, N = B * S, heads * 64
    dy, W, other = torch.randn(M, K), torch.randn(K, N), torch.randn(M, N)
    ref = dy.double() @ W.double()
    dx = torch.empty(M, N, device=DEV)
    out = torch.full((B, heads, S), float("nan"), device=DEV)
    ops.linear_dgrad(dev(dy), K, dev(W), N, dx, N, M, K, N, rowdot=(dev(other), out, S))
    assert _rel(dx, ref) < 2e-6
    want = (ref.view(B, S, heads, 64) * other.double().view(B, S, heads, 64)).sum(-1).permute(0, 2, 1)
    assert _rel(out, want) < 5e-6
    with pytest.raises(RuntimeError):        # partial tiles are refused, the caller keeps the separate delta kernel
        ops.linear_dgrad(dev(dy[:100]), K, dev(W), N, dx[:100], N, 100, K, N, rowdot=(dev(other[:100]), out, 50))


@pytest.mark.parametrize("rows,N,K", [(5000, 384, 256), (777, 1041, 256), (4100, 512, 320), (4500, 256, 512), (4200, 1041, 256)])
def test_gemm_wgrad_splitk(rows, N, K, gemm_mode):
    ldy, ldx = (N + 3) // 4 * 4, K
    dy = torch.zeros(rows, ldy); dy[:, :N] = torch.randn(rows, N)
    x = torch.randn(rows, ldx)
    dW = torch.ones(N, K, device=DEV)
    ops.linear_wgrad(dev(dy), ldy, dev(x), ldx, dW, K, rows, N, K)
    ref = 1.0 + dy[:, :N].double().T @ x.double()
    e = _rel(dW, ref)
    _record(f"gemm_wgrad_{rows}x{N}x{K}_{gemm_mode}", e)
    assert e < 5e-6


def test_gemm_wgrad_deterministic_form(gemm_mode):
    """gamer_gemm_desc.wgrad_ws: the chunk partial tiles go to a workspace and are added in chunk order - the fp64 product at
    the atomics form's tolerance, accumulate semantics (C += ...), ragged tiles, expert segments with an empty expert, and
    the SAME BITS from repeated launches (which the atomics form does not promise)."""
    torch.manual_seed(12)
    with ops.deterministic(True):
        for rows, N, K in ((5000, 384, 256), (777, 1041, 256), (4100, 512, 320)):
            dy, x = torch.randn(rows, N + (-N) % 4), torch.randn(rows, K)
            dy[:, N:] = 0
            outs = []
            for rep in range(3):
                dW = torch.ones(N, K, device=DEV)
                ops.linear_wgrad(dev(dy), dy.shape[1], dev(x), K, dW, K, rows, N, K, kchunk=256 if rows < 1000 else 512)
                outs.append(dW)
            assert _rel(outs[0], 1.0 + dy[:, :N].double().T @ x.double()) < 5e-6
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        E, Din, I = 6, 320, 512
        sizes = [0, 700, 129, 1, 300, 128]
        T = sum(sizes)
        offs = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
        x, dy = torch.randn(T, Din), torch.randn(T, I)
        outs = []
        for rep in range(2):
            dW = torch.zeros(E, I, Din, device=DEV)
            ops.linear_wgrad(dev(dy), I, dev(x), Din, dW, Din, T, I, Din, groups=E, group_offsets=dev(offs), strideC=I * Din,
                             kchunk=256)
            outs.append(dW)
        refw = torch.zeros(E, I, Din, dtype=torch.float64)
        for e in range(E):
            a, b = int(offs[e]), int(offs[e + 1])
            refw[e] = dy[a:b].double().T @ x[a:b].double()
        assert _rel(outs[0], refw) < 5e-6 and torch.equal(outs[0], outs[1]) and float(outs[0][0].abs().max()) == 0.0


@pytest.mark.parametrize("T", [1000, 4224])
@pytest.mark.parametrize("p_drop", [0.0, 0.2])
def test_gemm_fused_residual_epilogue(p_drop, gemm_mode, T):
    """C[map(m)] = resid[map(m)] + dropout(x W^T): same values and the same mask as GEMM followed by the
    stand-alone residual kernel (plain and expert-grouped with the sorted-slot -> token map)."""
    N, K, seed = 256, 384, 4242
    x, W, resid = torch.randn(T, K), torch.randn(N, K) * 0.1, torch.randn(T, N)
    y = torch.empty(T, N, device=DEV)
    ops.linear_fwd(dev(x), K, dev(W), K, y, N, T, N, K)
    ref = dev(resid.clone())
    ops.residual_dropout_fwd(ref, y, p_drop, seed)
    out = torch.full((T, N), 7.0, device=DEV)
    ops.gemm(dev(x), K, 1, dev(W), K, 1, out, N, T, N, K, resid=dev(resid), p_drop=p_drop, seed=seed)
    assert float((out - ref).abs().max()) < 1e-5
    # grouped + row map (the expert down projection): rows in sorted order, output scattered to tokens
    E, I = 6, 512
    sizes = [0, 300, 129, 1, 300, 270] if T == 1000 else [0, 1300, 129, 1, 1524, 1270]
    offs = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    perm = torch.randperm(T).int()                   # sorted slot -> token
    slot = torch.empty(T, dtype=torch.int32); slot[perm.long()] = torch.arange(T, dtype=torch.int32)
    hm, Wd = torch.randn(T, I), torch.randn(E, N, I) * 0.05
    yd = torch.empty(T, N, device=DEV)
    ops.linear_fwd(dev(hm), I, dev(Wd), I, yd, N, T, N, I, groups=E, group_offsets=dev(offs), strideB=N * I)
    ref2 = dev(resid.clone())
    ops.residual_dropout_fwd(ref2, yd, p_drop, seed, dev(slot))
    out2 = torch.full((T, N), 7.0, device=DEV)
    ops.gemm(dev(hm), I, 1, dev(Wd), I, 1, out2, N, T, N, I, groups=E, group_offsets=dev(offs), strideB=N * I,
             resid=dev(resid), row_map=dev(perm), p_drop=p_drop, seed=seed)
    assert float((out2 - ref2).abs().max()) < 1e-5


@pytest.mark.parametrize("scale", [1, 5])
def test_gemm_grouped_experts(gemm_mode, scale):
    E, Din, I = 6, 320, 512
    sizes = [scale * n for n in [0, 700, 129, 1, 300, 128]]     # an empty expert, ragged and exact tiles
    T = sum(sizes)
    offs = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    x = torch.randn(T, Din)
    W = torch.randn(E, I, Din) * 0.1
    y = torch.full((T, I), 9.0, device=DEV)
    ops.linear_fwd(dev(x), Din, dev(W), Din, y, I, T, I, Din, groups=E, group_offsets=dev(offs), strideB=I * Din)
    ref = torch.empty(T, I, dtype=torch.float64)
    for e in range(E):
        a, b = int(offs[e]), int(offs[e + 1])
        ref[a:b] = x[a:b].double() @ W[e].double().T
    e1 = _rel(y, ref)
    # grouped dgrad
    dy = torch.randn(T, I)
    dx = torch.zeros(T, Din, device=DEV)
    ops.linear_dgrad(dev(dy), I, dev(W), Din, dx, Din, T, I, Din, groups=E, group_offsets=dev(offs), strideB=I * Din)
    refd = torch.empty(T, Din, dtype=torch.float64)
    for e in range(E):
        a, b = int(offs[e]), int(offs[e + 1])
        refd[a:b] = dy[a:b].double() @ W[e].double()
    e2 = _rel(dx, refd)
    # grouped wgrad
    dW = torch.zeros(E, I, Din, device=DEV)
    ops.linear_wgrad(dev(dy), I, dev(x), Din, dW, Din, T, I, Din, groups=E, group_offsets=dev(offs),
                     strideC=I * Din, kchunk=256)
    refw = torch.zeros(E, I, Din, dtype=torch.float64)
    for e in range(E):
        a, b = int(offs[e]), int(offs[e + 1])
        refw[e] = dy[a:b].double().T @ x[a:b].double()
    e3 = _rel(dW, refw)
    _record("gemm_grouped_" + gemm_mode, [e1, e2, e3])
    assert e1 < 2e-6 and e2 < 2e-6 and e3 < 5e-6
    assert float(dW[0].abs().max()) == 0.0


@pytest.mark.parametrize("B,S", [(2, 128), (16, 40), (128, 5)])
@pytest.mark.parametrize("cross", [False, True])
def test_gemm_qkv_epilogue_equals_gemm_then_qknorm_rope(cross, gemm_mode, B, S):
    """The q|k|v projection with the per-head RMSNorm + RoPE epilogue (gamer_gemm_desc.qk_*) against the two-kernel form:
    the same raw q|k|v (with the behaviour biases in the cross attention) and the same rotated q / k.  S = 40 and S = 5:
    a wave's 64-row patch spans several sequences (the RoPE position wraps more than once per patch)."""
    nq, nkv, H, nb1 = 6, 3, 256, 4
    T, QKV = B * S, (6 + 2 * 3) * 64
    g = torch.Generator().manual_seed(17)
    x, W = torch.randn(T, H, generator=g), torch.randn(QKV, H, generator=g) * 0.1
    wq, wk = 1 + 0.1 * torch.randn(64, generator=g), 1 + 0.1 * torch.randn(64, generator=g)
    cos, sin = orc.rope_tables(S, 64, 1e6)
    pos_ids = dev(torch.randint(0, S, (T,), generator=g).int()) if cross else None          # session-style positions
    bias = dict(bias_q=dev(torch.randn(nb1, nq * 64, generator=g)), bias_k=dev(torch.randn(nb1, nkv * 64, generator=g)),
                bias_v=dev(torch.randn(nb1, nkv * 64, generator=g)),
                act_idx=dev(torch.randint(0, nb1, (T,), generator=g).int())) if cross else {}
    ref = torch.empty(T, QKV, device=DEV)
    q_ref, k_ref = torch.empty(T, nq * 64, device=DEV), torch.empty(T, nkv * 64, device=DEV)
    ops.linear_fwd(dev(x), H, dev(W), H, ref, QKV, T, QKV, H)
    ops.qknorm_rope_fwd(ref, S, nq, nkv, dev(wq), dev(wk), 1e-6, dev(cos), dev(sin), q_ref, k_ref, pos_ids=pos_ids, **bias)
    out = torch.full((T, QKV), float("nan"), device=DEV)
    q_rot, k_rot = torch.full_like(q_ref, float("nan")), torch.full_like(k_ref, float("nan"))
    assert ops.qkv_fused_ok(dev(x), T, QKV)
    ops.gemm(dev(x), H, 1, dev(W), H, 1, out, QKV, T, QKV, H,
             qknorm=dict(wq=dev(wq), wk=dev(wk), eps=1e-6, cos=dev(cos), sin=dev(sin), q_rot=q_rot, k_rot=k_rot, pos_ids=pos_ids,
                         S=S, nq=nq, nkv=nkv, **bias))
    assert torch.equal(out, ref)
    assert float((q_rot - q_ref).abs().max()) < 2e-6 * float(q_ref.abs().max())
    assert float((k_rot - k_ref).abs().max()) < 2e-6 * float(k_ref.abs().max())
    with pytest.raises(RuntimeError):          # partial row tiles keep the two-kernel form
        ops.gemm(dev(x[:100]), H, 1, dev(W), H, 1, out[:100], QKV, 100, QKV, H,
                 qknorm=dict(wq=dev(wq), wk=dev(wk), eps=1e-6, cos=dev(cos), sin=dev(sin), q_rot=q_rot, k_rot=k_rot, S=S, nq=nq,
                             nkv=nkv))


@pytest.mark.parametrize("terms", [6, 9])
def test_gemm_split_with_precut_weight_planes_is_bit_identical(terms):
    """gamer_split3_planes + gamer_gemm_desc.b_planes: the B operand (weights) cut once into three bf16 planes instead of
    in every row tile.  Forward (k-contiguous B), input gradient (row-contiguous B), grouped experts (strideB), residual
    epilogue; an edge shape (partial tiles ignore the planes).  Same bits as the in-kernel cut, and the planes ARE the cut:
    p0 + p1 + p2 == x exactly."""
    g = torch.Generator().manual_seed(9)
    E, I, din, T = 3, 256, 128, 640
    flat = (torch.randn(E * I * din + 384 * din, generator=g) * torch.exp2(torch.randint(-12, 8, (E * I * din + 384 * din,), generator=g).float())).to(DEV)
    planes = torch.empty(3, flat.numel(), dtype=torch.bfloat16, device=DEV)
    ops.split3_planes(flat, planes)
    assert torch.equal(planes[0].float() + planes[1].float() + planes[2].float(), flat)       # exact in fp32: 24 bits in three pieces
    Wexp = flat[:E * I * din].view(E * I, din)
    W2 = flat[E * I * din:].view(384, din)
    x = dev(torch.randn(T, din, generator=g))
    dy = dev(torch.randn(T, 384, generator=g))
    offs = dev(torch.tensor([0, 256, 384, 640], dtype=torch.int32))
    res = dev(torch.randn(T, 384, generator=g))

    def run():
        y = torch.empty(T, 384, device=DEV); ops.linear_fwd(x, din, W2, din, y, 384, T, 384, din)
        yg = torch.empty(T, I, device=DEV); ops.linear_fwd(x, din, Wexp, din, yg, I, T, I, din, groups=E, group_offsets=offs, strideB=I * din)
        dx = torch.empty(T, din, device=DEV); ops.linear_dgrad(dy, 384, W2, din, dx, din, T, 384, din)
        yr = torch.empty(T, 384, device=DEV); ops.gemm(x, din, 1, W2, din, 1, yr, 384, T, 384, din, resid=res, p_drop=0.2, seed=5)
        ye = torch.empty(100, 384, device=DEV); ops.linear_fwd(x[:100], din, W2, din, ye, 384, 100, 384, din)     # partial row tile
        return [t.clone() for t in (y, yg, dx, yr, ye)]
    with ops.f32_matmul(terms):
        plain = run()
    with ops.f32_matmul(terms, (flat.data_ptr(), flat.numel() * 4, planes.data_ptr(), planes.stride(0))):
        pre = run()
    for a, b in zip(plain, pre):
        assert torch.equal(a, b)
    ref = x.double() @ W2.double().T
    assert float((pre[0].double() - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("terms", [6, 9])
def test_gemm_split_is_exact_where_fp32_is(terms):
    """The three bf16 pieces carry all 24 bits of an operand: (a) small integers - every product and partial sum is exact
    in fp32 - give bit-identical results to the fp32 MFMA on all three layouts; (b) values that need the last mantissa
    bit (1 + 2^-23, 1 - 2^-24, 2^-120 * odd) times powers of two come back exactly; (c) with BOTH operands using their
    low bits the nine-product form is exact where the sum is representable, and the six-product form is off by at
    most the three omitted piece products (< 3 * 2^-24 per term)."""
    torch.manual_seed(5)
    M, N, K = 384, 256, 96
    x = torch.randint(-8, 9, (M, K)).float()
    w = torch.randint(-8, 9, (N, K)).float()
    dy = torch.randint(-8, 9, (M, N)).float()
    outs = {}
    for mode in (0, terms):
        ops.set_f32_matmul(mode)
        y, dx, dW = torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV), torch.zeros(N, K, device=DEV)
        ops.linear_fwd(dev(x), K, dev(w), K, y, N, M, N, K)
        ops.linear_dgrad(dev(dy), N, dev(w), K, dx, K, M, N, K)
        ops.linear_wgrad(dev(dy), N, dev(x), K, dW, K, M, N, K)
        outs[mode] = (y.cpu(), dx.cpu(), dW.cpu())
    ops.set_f32_matmul(0)
    for a, b in zip(outs[0], outs[terms]):
        assert torch.equal(a, b)
    assert torch.equal(outs[terms][0].double(), x.double() @ w.double().T)
    # (b) one non-zero product per output element: nothing to round
    vals = torch.tensor([1 + 2.0 ** -23, 1 - 2.0 ** -24, 3 * 2.0 ** -120, -(2.0 ** 20 + 1), 1.9999998807907104, 7 * 2.0 ** 100])
    xb = torch.zeros(128, 32); xb[torch.arange(128), torch.arange(128) % 32] = vals[torch.arange(128) % 6]
    wb = torch.zeros(128, 32); wb[torch.arange(128), torch.arange(128) % 32] = 2.0 ** (torch.arange(128) % 5 - 2).float()
    ops.set_f32_matmul(terms)
    yb = torch.empty(128, 128, device=DEV)
    ops.linear_fwd(dev(xb), 32, dev(wb), 32, yb, 128, 128, 128, 32)
    ops.set_f32_matmul(0)
    assert torch.equal(yb.cpu().double(), xb.double() @ wb.double().T)
    # (c) both operands with low bits: a = 1 + 2^-12 + 2^-23 (all three pieces non-zero), b likewise
    a, b = 1 + 2.0 ** -12 + 2.0 ** -23, 1 - 2.0 ** -11 + 2.0 ** -22
    xc = torch.zeros(128, 32); xc[:, 0] = a
    wc = torch.zeros(128, 32); wc[:, 0] = b
    ops.set_f32_matmul(terms)
    yc = torch.empty(128, 128, device=DEV)
    ops.linear_fwd(dev(xc), 32, dev(wc), 32, yc, 128, 128, 128, 32)
    ops.set_f32_matmul(0)
    exact = float(torch.tensor(a, dtype=torch.float64) * torch.tensor(b, dtype=torch.float64))
    err = float((yc.cpu().double() - exact).abs().max())
    assert err <= (2.0 ** -24 if terms == 9 else 4 * 2.0 ** -24) * abs(exact)


def test_gemm_split_error_is_not_larger_than_the_fp32_mfma():
    """Wide dynamic range inputs at a train-step shape: error against the fp64 product of the same fp32 inputs, relative
    to sum_k |a_k b_k|, for the fp32 MFMA and both bf16-piece forms."""
    torch.manual_seed(11)
    T, N, K = 4096, 768, 256
    x = torch.randn(T, K) * torch.exp(torch.randn(T, K))
    W = torch.randn(N, K) * 0.05
    ref, sc = x.double() @ W.double().T, x.double().abs() @ W.double().abs().T
    errs = {}
    for mode in ("f32", "split3", "split6", "split9"):
        ops.set_f32_matmul(mode)
        y = torch.empty(T, N, device=DEV)
        ops.linear_fwd(dev(x), K, dev(W), K, y, N, T, N, K)
        e = (y.cpu().double() - ref).abs() / sc
        errs[mode] = (float(e.max()), float(e.pow(2).mean().sqrt()))
    ops.set_f32_matmul("f32")
    _record("gemm_split_error", errs)
    for mode in ("split3", "split6", "split9"):
        assert errs[mode][0] < 1.5 * errs["f32"][0] + 1e-8 and errs[mode][1] < 1.25 * errs["f32"][1]
        assert errs[mode][0] < 2e-6


def test_gemm_split3_weight_planes_are_bit_identical():
    """matmul="split3": from the second pass on the parameters' maxima come from one launch and their fp16 piece planes are
    built once per pass (gamer_absmax_multi_f32 + gamer_split2h_planes_multi); forward, input-gradient and grouped (expert)
    GEMMs that read the planes give the bits of the in-kernel cut."""
    torch.manual_seed(6)
    T, N, K, E = 1024, 384, 256, 3
    flat = dev(torch.randn(N * K + E * N * K + 8) * 0.05)
    W = flat[:N * K].view(N, K)
    WE = flat[N * K:N * K + E * N * K].view(E * N, K)
    x, dy = dev(torch.randn(T, K) * 3), dev(torch.randn(T, N) * 1e-3)
    offs = dev(torch.tensor([0, 256, 640, T], dtype=torch.int32))
    cache = ops.amax_reuse()
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    outs = []
    with ops.f32_matmul("split3"), cache:
        for it in range(2):
            cache.reset()
            y, dx, ye = torch.empty(T, N, device=DEV), torch.empty(T, K, device=DEV), torch.empty(T, N, device=DEV)
            ops.linear_fwd(x, K, W, K, y, N, T, N, K)
            ops.linear_dgrad(dy, N, W, K, dx, K, T, N, K)
            ops.linear_fwd(x, K, WE, K, ye, N, T, N, K, strideB=N * K, groups=E, group_offsets=offs)
            outs.append((y.clone(), dx.clone(), ye.clone()))
            assert (len(cache._plane_keys) > 0) == (it == 1)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert _rel(outs[1][0], x.double().cpu() @ W.double().cpu().T) < 1e-6


@pytest.mark.parametrize("mag", [1e-30, 1e-9, 1e-4, 1.0, 1e6, 1e25])
def test_gemm_split3_error_does_not_depend_on_the_magnitude(mag):
    """The two-way fp16 form scales every operand tensor by a power of two from its own largest magnitude: gradients of
    1e-9 and activations of 1e6 come out at the same relative error (forward, dgrad and split-K wgrad layouts), small
    integers exactly, an all-zero operand gives zeros, and an Inf operand does not turn finite outputs of OTHER rows into
    garbage silently (it gives NaN / Inf in its row)."""
    torch.manual_seed(3)
    M, N, K = 640, 384, 160
    x = torch.randn(M, K) * torch.exp(torch.randn(M, K)) * mag
    w = torch.randn(N, K) * 0.05
    dy = torch.randn(M, N) * (1.0 / max(mag, 1e-30) if mag > 1 else 1.0) * 1e-3
    with ops.f32_matmul("split3"):
        y, dx, dW = torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV), torch.zeros(N, K, device=DEV)
        ops.linear_fwd(dev(x), K, dev(w), K, y, N, M, N, K)
        ops.linear_dgrad(dev(dy), N, dev(w), K, dx, K, M, N, K)
        ops.linear_wgrad(dev(dy), N, dev(x), K, dW, K, M, N, K)
    xd, wd, dyd = x.double(), w.double(), dy.double()
    for got, ref, sc in ((y, xd @ wd.T, xd.abs() @ wd.abs().T), (dx, dyd @ wd, dyd.abs() @ wd.abs()),
                         (dW, dyd.T @ xd, dyd.abs().T @ xd.abs())):
        e = ((got.cpu().double() - ref).abs() / sc.clamp_min(1e-300))
        assert float(e.max()) < 1e-6 and float(e.pow(2).mean().sqrt()) < 6e-8, (mag, float(e.max()))


def test_gemm_split3_integers_zeros_and_inf():
    torch.manual_seed(4)
    M, N, K = 256, 128, 64
    x = torch.randint(-30, 31, (M, K)).float()
    w = torch.randint(-30, 31, (N, K)).float()
    with ops.f32_matmul("split3"):
        y = torch.empty(M, N, device=DEV)
        ops.linear_fwd(dev(x), K, dev(w), K, y, N, M, N, K)
        assert torch.equal(y.cpu().double(), x.double() @ w.double().T)
        z = torch.full((M, N), 5.0, device=DEV)
        ops.linear_fwd(dev(torch.zeros(M, K)), K, dev(w), K, z, N, M, N, K)
        assert torch.equal(z.cpu(), torch.zeros(M, N))
        xi = x.clone(); xi[7, 3] = float("inf")
        yi = torch.empty(M, N, device=DEV)
        ops.linear_fwd(dev(xi), K, dev(w), K, yi, N, M, N, K)
        assert not torch.isfinite(yi[7].cpu()).all()
        ok = torch.ones(M, dtype=torch.bool); ok[7] = False
        assert torch.equal(yi.cpu()[ok].double(), (x.double() @ w.double().T)[ok])


# ----------------------------------------------------------------------------------------------
def _qknorm_ref(qkv, S, nq, nkv, wq, wk, cos, sin, bq=None, bk=None, bv=None, act=None):
    T = qkv.shape[0]
    B = T // S
    q = qkv[:, :nq * 64].view(B, S, nq, 64)
    k = qkv[:, nq * 64:(nq + nkv) * 64].view(B, S, nkv, 64)
    v = qkv[:, (nq + nkv) * 64:].view(B, S, nkv, 64)
    if bq is not None:
        q = q + bq[act].view(B, S, nq, 64)
        k = k + bk[act].view(B, S, nkv, 64)
        v = v + bv[act].view(B, S, nkv, 64)
    q = orc.apply_rope(orc.rmsnorm(q, wq, 1e-6), cos, sin)
    k = orc.apply_rope(orc.rmsnorm(k, wk, 1e-6), cos, sin)
    return q.reshape(T, -1), k.reshape(T, -1), v.reshape(T, -1)


@pytest.mark.parametrize("cross", [False, True])
def test_qknorm_rope_fwd_bwd(cross):
    B, S, nq, nkv, nb1 = 3, 35, 2, 1, 4
    T = B * S
    ld = (nq + 2 * nkv) * 64
    qkv = torch.randn(T, ld)
    wq, wk = 1 + 0.1 * torch.randn(64), 1 + 0.1 * torch.randn(64)
    cos, sin = orc.rope_tables(S, 64, 1e6)
    act = torch.randint(0, nb1, (T,))
    bq, bk, bv = torch.randn(nb1, nq * 64), torch.randn(nb1, nkv * 64), torch.randn(nb1, nkv * 64)
    leaves = [t.double().requires_grad_(True) for t in (qkv, wq, wk, bq, bk, bv)]
    args = (leaves[3], leaves[4], leaves[5], act.view(B, S)) if cross else ()
    qr, kr, vr = _qknorm_ref(leaves[0], S, nq, nkv, leaves[1], leaves[2], cos.double(), sin.double(), *args)
    d_qkv = dev(qkv.clone())
    q_rot = torch.empty(T, nq * 64, device=DEV)
    k_rot = torch.empty(T, nkv * 64, device=DEV)
    kw = dict(bias_q=dev(bq), bias_k=dev(bk), bias_v=dev(bv), act_idx=dev(act.int())) if cross else {}
    ops.qknorm_rope_fwd(d_qkv, S, nq, nkv, dev(wq), dev(wk), 1e-6, dev(cos), dev(sin), q_rot, k_rot, **kw)
    e = [_rel(q_rot, qr), _rel(k_rot, kr), _rel(d_qkv[:, (nq + nkv) * 64:], vr)]
    _record(f"qknorm_fwd_cross{int(cross)}", e)
    assert max(e) < 3e-6
    dq, dk, dv = torch.randn(T, nq * 64), torch.randn(T, nkv * 64), torch.randn(T, nkv * 64)
    (qr * dq.double()).sum().backward(retain_graph=True)
    (kr * dk.double()).sum().backward(retain_graph=True)
    (vr * dv.double()).sum().backward()
    dqkv = torch.zeros(T, ld, device=DEV)
    dqkv[:, (nq + nkv) * 64:] = dev(dv)
    dwq, dwk = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    if cross:
        dbq, dbk, dbv = (torch.zeros_like(dev(t)) for t in (bq, bk, bv))
        ops.qknorm_rope_bwd(d_qkv, dev(dq), dev(dk), S, nq, nkv, dev(wq), dev(wk), 1e-6, dev(cos), dev(sin), dqkv, dwq,
                            dwk, bias_q=dev(bq), bias_k=dev(bk), act_idx=dev(act.int()), nb1=nb1, dbias_q=dbq,
                            dbias_k=dbk, dbias_v=dbv)
    else:
        ops.qknorm_rope_bwd(d_qkv, dev(dq), dev(dk), S, nq, nkv, dev(wq), dev(wk), 1e-6, dev(cos), dev(sin), dqkv, dwq,
                            dwk)
    e = [_rel(dqkv, leaves[0].grad), _rel(dwq, leaves[1].grad), _rel(dwk, leaves[2].grad)]
    if cross:
        e += [_rel(dbq, leaves[3].grad), _rel(dbk, leaves[4].grad), _rel(dbv, leaves[5].grad)]
    _record(f"qknorm_bwd_cross{int(cross)}", e)
    assert max(e) < 2e-5


# ----------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, ok, nq, nkv, scale, mult=None):
    """Dense reference with the empty-row rule (q [B,S,nq,64] ...), fp64 autograd friendly."""
    B, S = q.shape[:2]
    rep = nq // nkv
    kq, vq = k.repeat_interleave(rep, 2), v.repeat_interleave(rep, 2)
    s = torch.einsum("bind,bjnd->bnij", q, kq) * scale
    empty = ~ok.any(-1)
    s_eff = torch.where(empty[:, None, :, None], s - s.detach(), s.masked_fill(~ok[:, None], float("-inf")))
    p = torch.softmax(s_eff, -1)
    lse = torch.logsumexp(s_eff, -1)
    pd = p if mult is None else p * mult
    return torch.einsum("bnij,bjnd->bind", pd, vq), lse, empty


def _run_attn(batch, cross, B, S, nq, nkv, p_drop=0.0, seed=1234, q=None, k=None, v=None, d_o=None, router=None,
              use_order=False, spill=False):
    T = B * S
    if router is None:
        router = ops.alloc_router_outputs(B, S, DEV)
        lut = torch.full((64,), -1, dtype=torch.int32)
        ops.router_fwd(dev(batch["input_ids"]), dev(batch["attention_mask"]), dev(batch["actions"]), dev(lut), 5, 4, 8,
                       router)
    kl = router["kl_cross"] if cross else router["kl_self"]
    ql = router["ql_cross"] if cross else None
    re_ = router["empty_cross"] if cross else router["empty_self"]
    te = router["tile_empty_cross"] if cross else router["tile_empty_self"]
    order = None
    if use_order:
        n_t = (S + 31) // 32
        order = (torch.empty(B, S, dtype=torch.int32, device=DEV), torch.empty(B, n_t, dtype=torch.int32, device=DEV),
                 torch.empty(B, n_t, dtype=torch.int32, device=DEV))
        ops.attn_row_order(re_, *order)
    ldv = (nq + 2 * nkv) * 64                       # v lives inside a qkv buffer, as in the model
    qkv = torch.zeros(T, ldv, device=DEV)
    qkv[:, (nq + nkv) * 64:] = dev(v.reshape(T, -1))
    vview = qkv[:, (nq + nkv) * 64:]
    o = torch.empty(T, nq * 64, device=DEV)
    lse = torch.empty(B, nq, S, device=DEV)
    dq_, dk_ = dev(q.reshape(T, -1)), dev(k.reshape(T, -1))
    split = spill in ("split", "split_spill", "split_h2")   # the same attention with its products on the 16-bit pipe (gamer_attn_*_split)
    h2 = spill == "split_h2"                          # ... in the three-product fp16 form
    if split:
        ops.attn_fwd_split(dq_, nq * 64, dk_, nkv * 64, vview, ldv, kl, ql, re_, B, S, nq, nkv, 0.125, p_drop, seed, o, lse,
                           order=order, h2=h2)
    else:
        ops.attn_fwd(dq_, nq * 64, dk_, nkv * 64, vview, ldv, kl, ql, re_, te, B, S, nq, nkv, 0.125, p_drop, seed, o, lse,
                     order=order)
    res = dict(o=o, lse=lse, router=router, order=order)
    if d_o is not None:
        delta = torch.empty(B, nq, S, device=DEV)
        dq = torch.empty(T, nq * 64, device=DEV)
        dk = torch.empty(T, nkv * 64, device=DEV)
        dqkv = torch.zeros(T, ldv, device=DEV)
        dvv = dqkv[:, (nq + nkv) * 64:]
        if split:
            ds_work = (torch.full((ops.attn_ds_work_numel(B, S, nq),), float("nan"), device=DEV)
                       if spill == "split_spill" else None)
            ops.attn_bwd_split(dq_, nq * 64, dk_, nkv * 64, vview, ldv, o, dev(d_o.reshape(T, -1)), lse, kl, ql, re_, te, B, S,
                               nq, nkv, 0.125, p_drop, seed, delta, dq, nq * 64, dk, nkv * 64, dvv, ldv, order=order,
                               ds_work=ds_work, h2=h2, dv_of=dqkv if h2 else None)
            res["dqkv"] = dqkv
        else:
            ds_work = torch.full((ops.attn_ds_work_numel(B, S, nq),), float("nan"), device=DEV) if spill else None
            ops.attn_bwd(dq_, nq * 64, dk_, nkv * 64, vview, ldv, o, dev(d_o.reshape(T, -1)), lse, kl, ql, re_, te, B, S, nq,
                         nkv, 0.125, p_drop, seed, delta, dq, nq * 64, dk, nkv * 64, dvv, ldv, order=order, ds_work=ds_work)
        res.update(dq=dq, dk=dk, dv=dvv)
    return res


@pytest.mark.parametrize("spill", [False, True, "split", "split_spill", "split_h2"])
@pytest.mark.parametrize("use_order", [False, True])
@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("n_items,B,nq,nkv", [(7, 3, 2, 1), (14, 2, 2, 1), (20, 2, 2, 1), (101, 2, 2, 1), (101, 11, 6, 3),
                                              (33, 40, 6, 3), (40, 3, 3, 3)])
def test_attention_fwd_bwd(cross, n_items, B, nq, nkv, use_order, spill):
    """The last two shapes give every persistent workgroup several (pair, tile) items, the regime the
    train step runs in (LDS reuse across items and tiles).  S = 70 leaves a 32-row wave tile entirely past the
    end of the sequence: its wave has no work and runs ahead of the others (caught an LDS reuse race)."""
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=7 + n_items, pad_rows={0: max(1, n_items // 3)})
    S = batch["input_ids"].shape[1]
    g = torch.Generator().manual_seed(n_items)
    q = torch.randn(B, S, nq, 64, generator=g)
    k = torch.randn(B, S, nkv, 64, generator=g)
    v = torch.randn(B, S, nkv, 64, generator=g)
    d_o = torch.randn(B, S, nq, 64, generator=g)
    self_ok, cross_ok = orc.mask_predicates(batch["attention_mask"], batch["actions"])
    ok = cross_ok if cross else self_ok
    leaves = [t.double().requires_grad_(True) for t in (q, k, v)]
    o_ref, lse_ref, empty = _attn_ref(*leaves, ok, nq, nkv, 0.125)
    (o_ref * d_o.double()).sum().backward()
    res = _run_attn(batch, cross, B, S, nq, nkv, q=q, k=k, v=v, d_o=d_o, use_order=use_order, spill=spill)
    if use_order:
        # the order is a stable partition: normal rows ascending, then empty rows ascending
        perm = res["order"][0].cpu().long()
        em = (~ok.any(-1))
        for bb in range(B):
            ref_perm = torch.cat([torch.nonzero(~em[bb]).flatten(), torch.nonzero(em[bb]).flatten()])
            assert torch.equal(perm[bb], ref_perm)
    T = B * S
    e_o = _rel(res["o"], o_ref.reshape(T, -1))
    ne = ~empty
    lse_got = res["lse"].cpu().permute(0, 2, 1)[ne]
    e_l = float((lse_got.double() - lse_ref.permute(0, 2, 1)[ne]).abs().max())
    e_dq = _rel(res["dq"], leaves[0].grad.reshape(T, -1))
    e_dk = _rel(res["dk"], leaves[1].grad.reshape(T, -1))
    e_dv = _rel(res["dv"], leaves[2].grad.reshape(T, -1))
    _record(f"attn_cross{int(cross)}_S{S}_B{B}_h{nq}_ord{int(use_order)}_spill{spill}", dict(o=e_o, lse=e_l, dq=e_dq, dk=e_dk, dv=e_dv,
                                                   empty_rows=int(empty.sum())))
    if cross:
        assert int((empty & batch["attention_mask"].bool()).sum()) > 0, "fixture must contain empty rows"
    # tolerance: fp32 MFMA + __expf against an fp64 reference
    assert e_o < 2e-5 and e_l < 2e-5
    assert e_dq < 5e-5 and e_dk < 5e-5 and e_dv < 5e-5


def test_attention_left_padding_empty_self_rows():
    """Left padding (eval layout) makes the first self-attention rows empty: uniform over all S keys."""
    B, S, nq, nkv = 2, 40, 2, 1
    batch = synthetic.make_batch(B, 8, 8, 3, seed=3)
    am = batch["attention_mask"].clone()
    am[0, :15] = 0
    batch["attention_mask"] = am
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, S, n, 64, generator=g) for n in (nq, nkv, nkv))
    d_o = torch.randn(B, S, nq, 64, generator=g)
    self_ok, _ = orc.mask_predicates(am, batch["actions"])
    leaves = [t.double().requires_grad_(True) for t in (q, k, v)]
    o_ref, _, empty = _attn_ref(*leaves, self_ok, nq, nkv, 0.125)
    assert int(empty.sum()) == 15
    (o_ref * d_o.double()).sum().backward()
    T = B * S
    for form in (False, "split", "split_spill", "split_h2"):
        for use_order in (False, True):
            res = _run_attn(batch, False, B, S, nq, nkv, q=q, k=k, v=v, d_o=d_o, use_order=use_order, spill=form)
            assert _rel(res["o"], o_ref.reshape(B * S, -1)) < 2e-5
        assert _rel(res["o"], o_ref.reshape(T, -1)) < 2e-5
        assert _rel(res["dq"], leaves[0].grad.reshape(T, -1)) < 5e-5
        assert _rel(res["dk"], leaves[1].grad.reshape(T, -1)) < 5e-5
        assert _rel(res["dv"], leaves[2].grad.reshape(T, -1)) < 5e-5


@pytest.mark.parametrize("spill", [False, True, "split", "split_spill", "split_h2"])
@pytest.mark.parametrize("use_order", [False, True])
@pytest.mark.parametrize("cross", [False, True])
def test_attention_dropout_mask_consistent_fwd_bwd(cross, use_order, spill):
    """Recover the keep-mask with q=k=0 and V = one-hot(j) (S <= 64), then check forward and
    backward with that exact mask against the dense reference."""
    B, n_items, nq, nkv, p = 2, 12, 2, 1, 0.2
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=21, pad_rows={1: 3})
    S = batch["input_ids"].shape[1]
    assert S <= 64
    self_ok, cross_ok = orc.mask_predicates(batch["attention_mask"], batch["actions"])
    ok = cross_ok if cross else self_ok
    zq = torch.zeros(B, S, nq, 64)
    zk = torch.zeros(B, S, nkv, 64)
    eye = torch.zeros(B, S, nkv, 64)
    eye[:, torch.arange(S), 0, torch.arange(S)] = 1.0
    r0 = _run_attn(batch, cross, B, S, nq, nkv, p_drop=p, seed=99, q=zq, k=zk, v=eye, use_order=use_order)
    pd = r0["o"].cpu().view(B, S, nq, 64)[..., :S].permute(0, 2, 1, 3).double()     # [B,nq,S,S] = p~
    p_ref = _attn_ref(zq.double(), zk.double(), eye.double(), ok, nq, nkv, 0.125)
    empty = p_ref[2]
    cnt = ok.sum(-1).clamp_min(1).double()
    punif = torch.where(empty[:, :, None], torch.full_like(ok, 1.0 / S, dtype=torch.float64), ok.double() / cnt[..., None])
    mult = torch.where(punif[:, None] > 0, pd / punif[:, None].clamp_min(1e-30), torch.zeros_like(pd))
    vals = mult[(punif[:, None] > 0).expand_as(mult)]
    assert bool(((vals - 1.25).abs() < 1e-4).logical_or(vals.abs() < 1e-7).all()), "mask values must be 0 or 1/(1-p)"
    keep_rate = float((vals > 0).double().mean())
    _record(f"attn_dropout_keep_rate_cross{int(cross)}_ord{int(use_order)}", keep_rate)
    assert abs(keep_rate - 0.8) < 0.03
    mult = torch.where(mult > 0.5, torch.full_like(mult, 1.25), torch.zeros_like(mult))
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, S, n, 64, generator=g) for n in (nq, nkv, nkv))
    d_o = torch.randn(B, S, nq, 64, generator=g)
    leaves = [t.double().requires_grad_(True) for t in (q, k, v)]
    o_ref, _, _ = _attn_ref(*leaves, ok, nq, nkv, 0.125, mult=mult)
    (o_ref * d_o.double()).sum().backward()
    res = _run_attn(batch, cross, B, S, nq, nkv, p_drop=p, seed=99, q=q, k=k, v=v, d_o=d_o, use_order=use_order,
                    spill=spill)
    T = B * S
    e = [_rel(res["o"], o_ref.reshape(T, -1)), _rel(res["dq"], leaves[0].grad.reshape(T, -1)),
         _rel(res["dk"], leaves[1].grad.reshape(T, -1)), _rel(res["dv"], leaves[2].grad.reshape(T, -1))]
    _record(f"attn_dropout_cross{int(cross)}_ord{int(use_order)}", e)
    assert max(e) < 5e-5


# ----------------------------------------------------------------------------------------------
def test_elementwise_dropout_family():
    T, H, p, seed = 257, 256, 0.2, 77
    x0, delta = torch.randn(T, H), torch.randn(T, H)
    src = torch.randperm(T).int()
    # p = 0
    x = dev(x0.clone())
    ops.residual_dropout_fwd(x, dev(delta), 0.0, seed, dev(src))
    assert torch.allclose(x.cpu(), x0 + delta[src.long()], atol=1e-6)
    # mask extraction with ones
    ones = torch.ones(T, H)
    z = torch.zeros(T, H, device=DEV)
    ops.residual_dropout_fwd(z, dev(ones), p, seed, None)
    mask = z.cpu()
    assert bool(((mask == 0) | ((mask - 1.25).abs() < 1e-6)).all())
    assert abs(float((mask > 0).float().mean()) - 0.8) < 0.01
    x = dev(x0.clone())
    ops.residual_dropout_fwd(x, dev(delta), p, seed, dev(src))
    assert torch.allclose(x.cpu(), x0 + mask * delta[src.long()], atol=1e-6)
    dd = torch.zeros(T, H, device=DEV)
    dx = torch.randn(T, H)
    ops.residual_dropout_bwd(dev(dx), p, seed, dd, dev(src))
    ref = torch.zeros(T, H)
    ref[src.long()] = mask * dx
    assert torch.allclose(dd.cpu(), ref, atol=1e-6)
    # a different seed gives a different mask
    z2 = torch.zeros(T, H, device=DEV)
    ops.residual_dropout_fwd(z2, dev(ones), p, seed + 1, None)
    assert float((z2.cpu() != mask).float().mean()) > 0.2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,I,ld", [(300, 512, 1024), (37, 96, 200), (1, 4, 8)])
def test_swiglu_on_the_fused_gate_up_buffer_equals_the_contiguous_kernels(rows, I, ld, dtype):
    """gamer_swiglu_fwd_ld / _bwd_ld read gate and up from the column halves of one [T, ld] buffer (the output of the fused
    gate|up projection): bit-identical values and dropout masks to gamer_swiglu_fwd / _bwd on two contiguous arrays, padding
    columns (ld > 2 I) untouched; with the maxima sink armed, the whole-gradient maximum of the backward lands in one slot."""
    p, seed = 0.2, 11
    gen = torch.Generator().manual_seed(rows)
    g = torch.randn(rows, I, generator=gen).to(dtype)
    u = torch.randn(rows, I, generator=gen).to(dtype)
    dhm = torch.randn(rows, I, generator=gen).to(dtype)
    gu = torch.full((rows, ld), 7.0, dtype=dtype)
    gu[:, :I], gu[:, I:2 * I] = g, u
    gu_d = dev(gu)
    hm_a, hm_b = torch.empty(rows, I, dtype=dtype, device=DEV), torch.empty(rows, I, dtype=dtype, device=DEV)
    ops.swiglu_fwd(dev(g), dev(u), rows * I, p, seed, hm_a)
    ops.swiglu_fwd_ld(gu_d, ld, rows, I, p, seed, hm_b)
    assert torch.equal(hm_a, hm_b)
    assert torch.equal(gu_d.cpu(), gu)                       # the forward writes nothing into its input
    dg, du = dev(g.clone()), dev(u.clone())
    ops.swiglu_bwd(dg, du, dev(dhm), rows * I, p, seed)
    ops.swiglu_bwd_ld(gu_d, ld, rows, I, dev(dhm), p, seed)
    assert torch.equal(gu_d[:, :I], dg) and torch.equal(gu_d[:, I:2 * I], du)
    assert bool((gu_d[:, 2 * I:] == 7.0).all())
    if dtype == torch.float32 and ld == 2 * I:
        with ops.f32_matmul("split3"), ops.amax_reuse(everything=True) as cache:
            gu2 = dev(gu)
            ops.swiglu_bwd_ld(gu2, ld, rows, I, dev(dhm), p, seed)
            key = cache._key(gu2.data_ptr(), (1, 0, 1, rows * ld, rows * ld))
            assert key in cache.pending, "the producer did not open a slot"
            assert _slot_value(cache.pending[key]) == float(gu2.abs().max())


@pytest.mark.parametrize("T,offsets", [(1024, None), (700, [0, 0, 130, 131, 389, 389, 700]), (37, None)])
def test_gemm_swiglu_backward_epilogue_equals_dgrad_then_swiglu_bwd(T, offsets):
    """gamer_gemm_desc.sw_gu: the input gradient of the experts' down projection with the SwiGLU backward in its epilogue
    (d(hm) never stored) against the two-kernel path - plain and grouped (ragged expert segments, empty experts: edge tiles),
    dropout on, and the maxima slot of the fused gradient."""
    H, I, p, seed = 256, 512, 0.2, 13
    E = 6 if offsets is not None else 1
    gen = torch.Generator().manual_seed(T)
    t0 = dev(torch.randn(T, H, generator=gen) * 0.1)
    Wd = dev(torch.randn(E * H, I, generator=gen) * 0.05)
    gu0 = torch.randn(T, 2 * I, generator=gen)
    grp = {}
    if offsets is not None:
        grp = dict(groups=E, group_offsets=dev(torch.tensor(offsets, dtype=torch.int32)), strideB=H * I)
    with ops.f32_matmul("split3"), ops.amax_reuse(everything=True) as cache:
        dhm = torch.empty(T, I, device=DEV)
        ops.linear_dgrad(t0, H, Wd, I, dhm, I, T, H, I, **grp)
        want = dev(gu0)
        ops.swiglu_bwd_ld(want, 2 * I, T, I, dhm, p, seed)
        got = dev(gu0)
        dummy = torch.full((T, I), 5.0, device=DEV)
        ops.gemm(t0, H, 1, Wd, 1, I, dummy, I, T, I, H, p_drop=p, seed=seed, swiglu_bwd=(got, 2 * I), **grp)
        key = cache._key(got.data_ptr(), (1, 0, 1, T * 2 * I, T * 2 * I))
        assert key in cache.pending
        slot = _slot_value(cache.pending[key])
    assert bool((dummy == 5.0).all()), "C must not be written"
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-6 * scale, float((got - want).abs().max()) / scale
    assert slot == float(got.abs().max())
    # rows outside every expert segment (grouped form with gaps) keep their gate | up values
    if offsets is not None and offsets[0] > 0:
        assert torch.equal(got[:offsets[0]].cpu(), gu0[:offsets[0]])


@pytest.mark.parametrize("T,offsets", [(1024, None), (700, [0, 0, 130, 131, 389, 389, 700]), (37, None)])
def test_gemm_bf16_swiglu_backward_epilogue_is_bit_identical_to_the_two_kernels(T, offsets):
    """gamer_gemm_bf16_desc.sw_gu: d(hm) is rounded to bf16 in the tile exactly as the unfused GEMM stores it, so the fused
    epilogue must reproduce gamer_gemm_bf16 + gamer_swiglu_bwd_ld_bf16 bit for bit (plain, ragged expert segments)."""
    H, I, p, seed = 256, 512, 0.2, 13
    BF = torch.bfloat16
    E = 6 if offsets is not None else 1
    gen = torch.Generator().manual_seed(T + 1)
    t0 = dev((torch.randn(T, H, generator=gen) * 0.1).to(BF))
    WT = dev((torch.randn(E * I, H, generator=gen) * 0.05).to(BF))          # per expert: W_down^T [I, H]
    gu0 = torch.randn(T, 2 * I, generator=gen).to(BF)
    grp = {}
    if offsets is not None:
        grp = dict(groups=E, group_offsets=dev(torch.tensor(offsets, dtype=torch.int32)), strideB=H * I)
    dhm = torch.empty(T, I, dtype=BF, device=DEV)
    ops.linear_dgrad_t(t0, H, WT, H, dhm, I, T, H, I, **grp)
    want = dev(gu0)
    ops.swiglu_bwd_ld(want, 2 * I, T, I, dhm, p, seed)
    got = dev(gu0)
    dummy = torch.full((T, I), 5.0, dtype=BF, device=DEV)
    ops.linear_dgrad_t(t0, H, WT, H, dummy, I, T, H, I, p_drop=p, seed=seed, swiglu_bwd=(got, 2 * I), **grp)
    assert bool((dummy == 5.0).all()), "C must not be written"
    assert torch.equal(got, want)


def test_colsum_reduce_batched_equals_the_single_table_kernel():
    rows, cols, n = 2048, 256, 5
    gen = torch.Generator().manual_seed(0)
    part = dev(torch.randn(n + 1, rows, cols, generator=gen))
    outs_a = [dev(torch.randn(cols, generator=gen)) for _ in range(n)]
    outs_b = [o.clone() for o in outs_a]
    for i in range(n):
        ops.colsum_reduce(part[i], outs_a[i], accumulate=True)
    table = torch.tensor([o.data_ptr() for o in outs_b], dtype=torch.int64, device=DEV)
    ops.colsum_reduce_batched(part, n, table, accumulate=True)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    ops.colsum_reduce_batched(part, 2, table, accumulate=False)
    assert torch.equal(outs_b[1], part[1].sum(0, dtype=torch.float32)) or _rel(outs_b[1], part[1].double().sum(0)) < 1e-5
    assert torch.equal(outs_b[2], outs_a[2])                  # tables past n are not touched


def test_swiglu_and_gate():
    n_rows, I, p, seed = 300, 512, 0.2, 5
    g, u = torch.randn(n_rows, I), torch.randn(n_rows, I)
    hm = torch.empty(n_rows, I, device=DEV)
    ops.swiglu_fwd(dev(g), dev(u), g.numel(), 0.0, seed, hm)
    gr, ur = g.double().requires_grad_(True), u.double().requires_grad_(True)
    ref = torch.nn.functional.silu(gr) * ur
    assert _rel(hm, ref) < 2e-6
    # dropout mask via ones: silu(big)=big -> use g = large so silu(g)~g? simpler: u = ones, g = c with silu(c) known
    c = torch.full((n_rows, I), 2.0)
    hm2 = torch.empty(n_rows, I, device=DEV)
    ops.swiglu_fwd(dev(c), dev(torch.ones(n_rows, I)), g.numel(), p, seed, hm2)
    mask = (hm2.cpu() / float(torch.nn.functional.silu(torch.tensor(2.0)))).round(decimals=3)
    assert bool(((mask == 0) | (mask == 1.25)).all())
    ops.swiglu_fwd(dev(g), dev(u), g.numel(), p, seed, hm)
    assert _rel(hm, ref * mask.double()) < 2e-6
    dhm = torch.randn(n_rows, I)
    (ref * mask.double() * dhm.double()).sum().backward()
    dg, du = dev(g.clone()), dev(u.clone())
    ops.swiglu_bwd(dg, du, dev(dhm), g.numel(), p, seed)
    assert _rel(dg, gr.grad) < 1e-5 and _rel(du, ur.grad) < 1e-5
    # output gate
    a, gate, dout = torch.randn(n_rows, 256), torch.randn(n_rows, 256), torch.randn(n_rows, 256)
    ar, gr2 = a.double().requires_grad_(True), gate.double().requires_grad_(True)
    refo = ar * torch.nn.functional.silu(gr2)
    out = torch.empty(n_rows, 256, device=DEV)
    ops.silu_gate_fwd(dev(a), dev(gate), out)
    assert _rel(out, refo) < 2e-6
    (refo * dout.double()).sum().backward()
    da, dgate = torch.empty_like(out), torch.empty_like(out)
    ops.silu_gate_bwd(dev(a), dev(gate), dev(dout), da, dgate)
    assert _rel(da, ar.grad) < 1e-5 and _rel(dgate, gr2.grad) < 1e-5
    # fused residual add + dropout: the same result (and mask) as silu_gate_fwd followed by residual_dropout_fwd,
    # and the backward equals residual_dropout_bwd followed by silu_gate_bwd
    x = torch.randn(n_rows, 256)
    two = torch.empty_like(out)
    ops.residual_dropout_fwd(dev(x), out, 0.2, 77, None, two)
    one = torch.empty_like(out)
    ops.silu_gate_fwd(dev(a), dev(gate), one, resid=dev(x), p=0.2, seed=77)
    assert torch.equal(one, two)
    masked = torch.empty_like(out)
    ops.residual_dropout_bwd(dev(dout), 0.2, 77, masked)
    da2, dg2 = torch.empty_like(out), torch.empty_like(out)
    ops.silu_gate_bwd(dev(a), dev(gate), masked, da2, dg2)
    da1, dg1 = torch.empty_like(out), torch.empty_like(out)
    ops.silu_gate_bwd(dev(a), dev(gate), dev(dout), da1, dg1, p=0.2, seed=77)
    assert torch.equal(da1, da2) and torch.equal(dg1, dg2)


def _slot_value(ptr_):
    """max over the 16 words of a maximum slot (ops.AMAX_WORDS apart by 16), as a float"""
    import ctypes
    words = torch.empty(ops.AMAX_WORDS, dtype=torch.int32, device=DEV)
    assert torch.cuda.current_stream().cuda_stream is not None
    # device-to-device copy of the slot through a tensor view of the pool that owns it
    for pool in ops._AMAX_REUSE.pools:
        base = pool.data_ptr()
        if base <= ptr_ < base + pool.numel() * 4:
            off = (ptr_ - base) // 4
            words.copy_(pool[off:off + ops.AMAX_WORDS])
            break
    else:
        raise AssertionError("slot not inside a pool")
    return float(words.cpu().view(torch.float32).max())


def test_amax_sinks_equal_the_maximum_of_what_the_kernel_wrote():
    """matmul="split3": every kernel that writes a GEMM / attention operand leaves max |value| in the slot the consumer will
    read (gamer_amax_sink) - bit-equal to the maximum of the tensor it wrote; gamer_absmax_f32 (dense and strided) and
    gamer_absmax_multi_f32 give the same."""
    torch.manual_seed(9)
    T, H, I, S, nq, nkv = 640, 256, 512, 64, 2, 1
    with ops.f32_matmul("split3"), ops.amax_reuse(everything=True) as cache:
        def pending(t, geom):
            key = cache._key(t.data_ptr(), geom)
            assert key in cache.pending, "the producer did not open a slot"
            return _slot_value(cache.pending[key])

        # RMSNorm forward (+ the behaviour-table columns of an injecting layer)
        x, w = dev(torch.randn(T, H) * 3), dev(1 + 0.1 * torch.randn(H))
        din = H + 64
        hin = torch.zeros(T, din, device=DEV)
        ops.rmsnorm_fwd(x, w, 1e-6, hin, din)
        tab, idx = dev(torch.randn(4, 64) * 9), dev(torch.randint(0, 4, (T,)).int())
        ops.rowtable_fwd(tab, idx, hin, din, H)
        assert pending(hin, (1, 0, T, din, din)) == float(hin.abs().max())
        # SwiGLU forward / backward, output gate backward
        g, u, hm = dev(torch.randn(T, I)), dev(torch.randn(T, I)), torch.empty(T, I, device=DEV)
        ops.swiglu_fwd(g, u, T * I, 0.2, 5, hm)
        assert pending(hm, (1, 0, 1, T * I, T * I)) == float(hm.abs().max())
        ops.swiglu_bwd(g, u, dev(torch.randn(T, I) * 1e-4), T * I, 0.2, 5)
        assert pending(g, (1, 0, 1, T * I, T * I)) == float(g.abs().max())
        assert pending(u, (1, 0, 1, T * I, T * I)) == float(u.abs().max())
        a, gate, dout = dev(torch.randn(T, H)), dev(torch.randn(T, H)), dev(torch.randn(T, H) * 1e-3)
        da, dgate = torch.empty(T, H, device=DEV), torch.empty(T, H, device=DEV)
        ops.silu_gate_bwd(a, gate, dout, da, dgate, p=0.2, seed=3)
        assert pending(da, (1, 0, 1, T * H, T * H)) == float(da.abs().max())
        assert pending(dgate, (1, 0, 1, T * H, T * H)) == float(dgate.abs().max())
        # RMSNorm backward: the masked copy for the next branch
        dx, part, mo = torch.zeros(T, H, device=DEV), torch.empty(512, H, device=DEV), torch.empty(T, H, device=DEV)
        ops.rmsnorm_bwd(x, w, dev(torch.randn(T, H) * 1e-5), H, 1e-6, dx, part, False, mask_out=mo, p=0.2, seed=11)
        assert pending(mo, (1, 0, T, H, H)) == float(mo.abs().max())
        # CE backward (the pad columns of the logits buffer do not count)
        B, V, ldl = T // S, 1041, 1056
        logits = dev(torch.randn(T, ldl) * 2)
        labels = dev(torch.randint(0, V, (B, S)))
        lse, rl, ls, cnt = (torch.zeros(n, device=DEV) for n in (T, T, 1, 1))
        ops.ce_fwd(logits, ldl, labels, V, 0.7, -100, lse, rl, ls, cnt)
        logits[:, V:] = 1e9
        ops.ce_bwd(logits, ldl, labels, V, 0.7, -100, lse, cnt, 0.0, 1.0)
        assert pending(logits, (1, 0, T, V, ldl)) == float(logits[:, :V].abs().max())
        # q/k norm + RoPE forward: q_rot, k_rot
        QKV = (nq + 2 * nkv) * 64
        qkv = dev(torch.randn(T, QKV) * 5)
        cos, sin = orc.rope_tables(S, 64, 1e6)
        q_rot, k_rot = torch.empty(T, nq * 64, device=DEV), torch.empty(T, nkv * 64, device=DEV)
        w64 = dev(1 + 0.1 * torch.randn(64))
        ops.qknorm_rope_fwd(qkv, S, nq, nkv, w64, w64, 1e-6, dev(cos), dev(sin), q_rot, k_rot)
        assert pending(q_rot, (1, 0, T, nq * 64, nq * 64)) == float(q_rot.abs().max())
        assert pending(k_rot, (1, 0, T, nkv * 64, nkv * 64)) == float(k_rot.abs().max())
        # ... with behaviour biases (the cross block) also the v columns it rewrites as v + bias_v (gamer_amax_sink3)
        qkv_c = dev(torch.randn(T, QKV) * 5)
        act = dev(torch.randint(0, 4, (T,)).int())
        bq, bk, bv = (dev(torch.randn(4, n * 64) * 3) for n in (nq, nkv, nkv))
        ops.qknorm_rope_fwd(qkv_c, S, nq, nkv, w64, w64, 1e-6, dev(cos), dev(sin), q_rot, k_rot, bias_q=bq, bias_k=bk, bias_v=bv,
                            act_idx=act)
        vc = qkv_c[:, (nq + nkv) * 64:]
        assert pending(vc, (1, 0, T, nkv * 64, QKV)) == float(vc.abs().max())
        assert pending(q_rot, (1, 0, T, nq * 64, nq * 64)) == float(q_rot.abs().max())
        # gamer_absmax_f32: dense, strided columns; gamer_absmax_multi_f32
        v = qkv[:, (nq + nkv) * 64:]
        assert _slot_value(ops.absmax_slot(v, 1, 0, T, nkv * 64, QKV)) == float(v.abs().max())
        assert _slot_value(ops.absmax_slot(qkv, 1, 0, T, QKV, QKV)) == float(qkv.abs().max())
        flat = dev(torch.randn(4096 + 8192 + 256))
        flat[5000] = -77.0
        table = torch.tensor([0, 4096, 4096, 8192, 12288, 256], dtype=torch.int64, device=DEV)
        first = cache._new_slot(flat.device)
        cache._new_slot(flat.device); cache._new_slot(flat.device)
        ops.call("gamer_absmax_multi_f32", flat.data_ptr(), table.data_ptr(), 3, first, ops.stream_ptr())
        for e, (o, n) in enumerate(((0, 4096), (4096, 8192), (12288, 256))):
            assert _slot_value(first + 4 * ops.AMAX_WORDS * e) == float(flat[o:o + n].abs().max())
        # a GEMM as a producer (gamer_gemm_desc.amax_c): the v columns of a q|k|v projection, full and ragged row counts, and dO with
        # the row-dot epilogue; a row-range-guarded tile (recomputed in fp32) reports the maximum of what it finally stored
        for Tg in (640, 300):
            xg, wg = dev(torch.randn(Tg, H)), dev(torch.randn(QKV, H) * 0.3)
            if Tg == 640:
                xg[128:256] *= 2.0 ** -20
            yg = torch.zeros(Tg, QKV, device=DEV)
            vg = yg[:, (nq + nkv) * 64:]
            ops.linear_fwd(xg, H, wg, H, yg, QKV, Tg, QKV, H, c_amax=(vg, (nq + nkv) * 64))
            assert pending(vg, (1, 0, Tg, nkv * 64, QKV)) == float(vg.abs().max())
        dyg, wo, other = dev(torch.randn(T, H) * 1e-3), dev(torch.randn(H, 128) * 0.1), dev(torch.randn(T, 128))
        dao, dl = torch.empty(T, 128, device=DEV), torch.empty(T // S, 2, S, device=DEV)
        ops.linear_dgrad(dyg, H, wo, 128, dao, 128, T, H, 128, rowdot=(other, dl, S), c_amax=(dao, 0))
        assert pending(dao, (1, 0, T, 128, 128)) == float(dao.abs().max())
        # the split attention as a producer: o (forward) and the v columns of d(q|k|v) (dK/dV kernel), self and row-ordered cross
        for cross in (False, True):
            cache.slots.clear()         # (everything=True keeps maxima by ADDRESS: new tensors may land where freed ones were)
            batch = synthetic.make_batch(3, 20, 8, 3, seed=13, pad_rows={0: 5})
            S2 = batch["input_ids"].shape[1]
            gq = torch.Generator().manual_seed(2)
            q4, k4, v4, do4 = (torch.randn(3, S2, n, 64, generator=gq) for n in (2, 1, 1, 2))
            res = _run_attn(batch, cross, 3, S2, 2, 1, p_drop=0.2, q=q4, k=k4, v=v4, d_o=do4 * 1e-4, use_order=cross, spill="split_h2")
            T2 = 3 * S2
            assert pending(res["o"], (1, 0, 1, T2 * 128, T2 * 128)) == float(res["o"].abs().max())
            assert pending(res["dqkv"], (1, 0, T2, 256, 256)) == float(res["dqkv"].abs().max())


@pytest.mark.parametrize("use_count", [True, False])
def test_cross_entropy_fwd_bwd(use_count):
    B, S, V, ldl, temp = 3, 35, 1041, 1056, 0.7
    logits = torch.randn(B, S, V) * 3
    labels = torch.randint(0, V, (B, S))
    labels[:, ::5] = -100
    labels[1, 20:] = -100
    buf = torch.zeros(B * S, ldl)
    buf[:, :V] = logits.view(-1, V)
    d_buf = dev(buf)
    lse = torch.empty(B * S, device=DEV)
    row_loss = torch.empty(B * S, device=DEV)
    loss_sum, count = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    ops.ce_fwd(d_buf, ldl, dev(labels), V, temp, -100, lse, row_loss, loss_sum, count)
    zr = (logits.double() / temp).requires_grad_(True)
    shift = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:]
    n_valid = int((shift != -100).sum())
    ref_sum = torch.nn.functional.cross_entropy(zr.view(-1, V), shift.reshape(-1), ignore_index=-100, reduction="sum")
    assert int(count.item()) == n_valid
    assert abs(float(loss_sum.item()) - float(ref_sum)) < 1e-4 * float(ref_sum)
    assert _rel(d_buf[:, :V], zr.view(-1, V)) < 1e-6          # logits scaled in place
    denom = float(n_valid) if use_count else 123.0
    (ref_sum / denom).backward()
    ops.ce_bwd(d_buf, ldl, dev(labels), V, temp, -100, lse, count if use_count else None, denom, 1.0)
    ref_g = zr.grad.view(-1, V) / temp                         # d loss / d (unscaled logits)
    e = _rel(d_buf[:, :V], ref_g)
    _record(f"ce_bwd_count{int(use_count)}", e)
    assert e < 1e-5


def test_clip_adamw_matches_hf_update():
    n, n_decay = 100003, 90000
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    m0 = torch.randn(n, generator=g) * 0.01
    v0 = torch.rand(n, generator=g) * 1e-4
    gr = torch.randn(n, generator=g) * 0.05
    pad = (4 - n % 4) % 4
    P_, G_, M_, V_ = (dev(torch.nn.functional.pad(t, (0, pad))) for t in (p0, gr, m0, v0))
    partial = torch.empty(256, device=DEV)
    norm = torch.zeros(1, device=DEV)
    ops.sumsq(G_, partial)
    ops.adamw(P_, G_, M_, V_, n_decay, 5e-4, 0.9, 0.999, 1e-8, 0.01, 3, 1.0, 1.0, partial, norm)
    params = {"a.weight": p0[:n_decay].clone(), "b_norm.weight": p0[n_decay:].clone()}
    grads = {"a.weight": gr[:n_decay].clone(), "b_norm.weight": gr[n_decay:].clone()}
    ms = {"a.weight": m0[:n_decay].clone(), "b_norm.weight": m0[n_decay:].clone()}
    vs = {"a.weight": v0[:n_decay].clone(), "b_norm.weight": v0[n_decay:].clone()}
    total = orc.clip_and_adamw(params, grads, ms, vs, step=3, lr=5e-4)
    assert abs(float(norm.item()) - float(total)) < 1e-4 * float(total)
    ref_p = torch.cat([params["a.weight"], params["b_norm.weight"]])
    assert _rel(P_[:n], ref_p) < 1e-6
    assert _rel(M_[:n], torch.cat([ms["a.weight"], ms["b_norm.weight"]])) < 1e-6
    assert _rel(V_[:n], torch.cat([vs["a.weight"], vs["b_norm.weight"]])) < 1e-6
