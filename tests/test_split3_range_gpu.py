"""Adversarial dynamic-range tests of the default fp32 form (matmul="split3": two fp16 pieces per value, ONE power-of-two
scale per operand tensor) and of its guards.

What per-tensor scaling gives up: an element 2^r below its tensor's largest magnitude keeps an absolute error of ~2^-38 of
that maximum, i.e. a ROW whose own maximum is 2^r below it is computed to ~2^(r - 38) of its own result.  Random data never
shows this (2^+-6 of range), so these tests build the cases that do - a block of low-magnitude rows (the input-gradient GEMM
of low-gradient tokens), one huge outlier element, peaked softmax scores, outliers in dO - and check

  * the GEMM guard (gamer_split3_guard, on by default): tiles holding such rows are recomputed in fp32 on the device; the tests
    FAIL with the guard off where the arithmetic says they must, pass with it on, and tiles the guard does not touch are
    bit-identical;
  * the attention backward's per-row scales of dO / dS in the dQ kernel (csrc/attention_split.hip);
  * what is deliberately NOT guarded, with the measured error next to the prediction: the weight operand, and the
    weight-gradient layout (its error stays relative to sum |a_k b_k| because the rows are the contraction index).

Errors are per element relative to sum_k |a_k b_k| of that element (so a small row is measured against its own size),
reduced per row where a test says "per row".  Reference: fp64 on the host.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

import os  # noqa: E402
import sys  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gamer_amd import ops  # noqa: E402
from test_ops_gpu import _attn_ref, _record, _run_attn, dev  # noqa: E402  (the same module object pytest collected)
from gamer_amd import synthetic  # noqa: E402
from oracle import qwen3multi_oracle as orc  # noqa: E402

DEV = "cuda"
FP32_BAR = 2e-6          # what the fp32 MFMA itself delivers on these shapes (measured 4e-7 .. 9e-7)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"


def _row_err(got, ref, scale):
    """max over the row of |got - ref| / sum_k |a_k b_k|  -> [rows]"""
    e = (got.detach().cpu().double() - ref).abs() / scale.clamp_min(1e-300)
    return e.max(dim=1).values


def _fwd_dgrad(x, w, dy):
    M, K = x.shape
    N = w.shape[0]
    y, dx = torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV)
    ops.linear_fwd(dev(x), K, dev(w), K, y, N, M, N, K)
    ops.linear_dgrad(dev(dy), N, dev(w), K, dx, K, M, N, K)
    return y, dx


def _refs(x, w, dy):
    xd, wd, dyd = x.double(), w.double(), dy.double()
    return (xd @ wd.T, xd.abs() @ wd.abs().T), (dyd @ wd, dyd.abs() @ wd.abs())


@pytest.mark.parametrize("M,N,K", [(512, 384, 256), (640, 320, 512), (300, 200, 96)])
def test_gemm_rows_far_below_the_tensor_maximum(M, N, K):
    """Blocks of rows 2^-18, 2^-22 and 2^-26 below the rest (forward: activations; input gradient: dY of low-gradient tokens)."""
    g = torch.Generator().manual_seed(M)
    x, w, dy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(M, N, generator=g)
    q = M // 4
    rows = torch.ones(M, 1)
    rows[q:2 * q], rows[2 * q:3 * q], rows[3 * q:] = 2.0 ** -18, 2.0 ** -22, 2.0 ** -26
    x, dy = x * rows, dy * rows
    (y_ref, y_sc), (dx_ref, dx_sc) = _refs(x, w, dy)
    with ops.f32_matmul("split3"):
        y1, dx1 = _fwd_dgrad(x, w, dy)
        with ops.split3_guard(False):
            y0, dx0 = _fwd_dgrad(x, w, dy)
    with ops.f32_matmul("f32"):
        yf, dxf = _fwd_dgrad(x, w, dy)
    for name, got1, got0, gotf, ref, sc in (("fwd", y1, y0, yf, y_ref, y_sc), ("dgrad", dx1, dx0, dxf, dx_ref, dx_sc)):
        e1, e0, ef = _row_err(got1, ref, sc), _row_err(got0, ref, sc), _row_err(gotf, ref, sc)
        _record(f"split3_rows_{name}_{M}x{N}x{K}", dict(guard=float(e1.max()), no_guard_by_block=[float(e0[i * q:(i + 1) * q].max()) for i in range(4)],
                                                         fp32_mfma=float(ef.max())))
        # with the guard every row is at the fp32 MFMA's level, measured against ITS OWN size
        assert float(e1.max()) < FP32_BAR, (name, float(e1.max()))
        # without it the 2^-26 block is where the arithmetic says: 2^(26 - 38) per element, ~1e-4 of the row's dot product
        assert float(e0[3 * q:].max()) > 10 * FP32_BAR, (name, float(e0[3 * q:].max()))
        # ... and the rows at full magnitude never needed it
        assert float(e0[:q].max()) < FP32_BAR
    # tiles without a flagged row are bit-identical with and without the guard (rows of the first full 128-row tile)
    if q >= 128:
        assert torch.equal(y1[:128], y0[:128]) and torch.equal(dx1[:128], dx0[:128])


@pytest.mark.parametrize("log2_outlier", [20, 30])
def test_gemm_one_outlier_in_the_activation_operand(log2_outlier):
    """One element 2^20 / 2^30 above everything else (a loss spike, a diverging activation): it sets the tensor's scale and
    pushes every other row 2^20 / 2^30 below the maximum."""
    torch.manual_seed(log2_outlier)
    M, N, K = 384, 256, 256
    x, w, dy = torch.randn(M, K), torch.randn(N, K) * 0.05, torch.randn(M, N) * 1e-3
    x[5, 7] *= 2.0 ** log2_outlier
    dy[200, 3] *= 2.0 ** log2_outlier
    (y_ref, y_sc), (dx_ref, dx_sc) = _refs(x, w, dy)
    with ops.f32_matmul("split3"):
        y1, dx1 = _fwd_dgrad(x, w, dy)
        with ops.split3_guard(False):
            y0, dx0 = _fwd_dgrad(x, w, dy)
    e1 = max(float(_row_err(y1, y_ref, y_sc).max()), float(_row_err(dx1, dx_ref, dx_sc).max()))
    e0 = max(float(_row_err(y0, y_ref, y_sc).max()), float(_row_err(dx0, dx_ref, dx_sc).max()))
    _record(f"split3_outlier_2^{log2_outlier}", dict(guard=e1, no_guard=e0))
    assert e1 < FP32_BAR, e1
    if log2_outlier == 30:
        assert e0 > 1e-4, e0          # 2^(30 - 38): every other row has 8 bits left without the guard
    else:
        assert e0 < 2e-5, e0          # 2^(20 - 38) = 4e-6: still below the attention bars, above the fp32 MFMA


def test_gemm_guard_with_fused_epilogues_and_expert_groups():
    """The fp32 recomputation feeds the same epilogues: residual + dropout-free scatter through a row map, expert segments
    (device offsets), the row-dot epilogue and accumulate."""
    torch.manual_seed(11)
    E, H, I = 6, 256, 512
    counts = torch.tensor([130, 0, 257, 128, 5, 384])
    offs = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).int()
    T = int(counts.sum())
    hm = torch.randn(T, I)
    hm[100:180] *= 2.0 ** -22                    # spans the boundary of expert 0 / expert 2 and two row tiles
    hm[700] *= 2.0 ** 25                         # and one outlier row further down
    Wd = torch.randn(E, H, I) * 0.05
    resid = torch.randn(T, H)
    perm = torch.randperm(T).int()
    ref = resid.double().clone()
    sc = torch.zeros(T, H, dtype=torch.float64)
    for e in range(E):
        r0, r1 = int(offs[e]), int(offs[e + 1])
        rows = perm[r0:r1].long()
        ref[rows] += hm[r0:r1].double() @ Wd[e].double().T
        sc[rows] = hm[r0:r1].double().abs() @ Wd[e].double().abs().T
    with ops.f32_matmul("split3"):
        out = torch.empty(T, H, device=DEV)
        ops.gemm(dev(hm), I, 1, dev(Wd.reshape(E * H, I)), I, 1, out, H, T, H, I, strideB=H * I, resid=dev(resid),
                 row_map=dev(perm), groups=E, group_offsets=dev(offs))
    # (the epilogue adds the product to an fp32 residual of magnitude ~1: half an ulp of the SUM is not the product's error)
    e = ((out.cpu().double() - ref).abs() - 2.0 ** -23 * ref.abs()).clamp_min(0) / sc.clamp_min(1e-300)
    live = sc.amax(1) > 0
    assert float(e[live].max()) < FP32_BAR, float(e[live].max())
    # and the product itself, without the residual, through the same grouped launch
    with ops.f32_matmul("split3"):
        prod = torch.empty(T, H, device=DEV)
        ops.gemm(dev(hm), I, 1, dev(Wd.reshape(E * H, I)), I, 1, prod, H, T, H, I, strideB=H * I, groups=E, group_offsets=dev(offs))
    pref = torch.cat([hm[int(offs[e_]):int(offs[e_ + 1])].double() @ Wd[e_].double().T for e_ in range(E)])
    psc = torch.cat([hm[int(offs[e_]):int(offs[e_ + 1])].double().abs() @ Wd[e_].double().abs().T for e_ in range(E)])
    assert float(_row_err(prod, pref, psc).max()) < FP32_BAR
    # row-dot epilogue (delta = dO . O of the attention backward) on a flagged tile, and accumulate
    M, N, K, S = 256, 128, 256, 64
    a, w2, other = torch.randn(M, K), torch.randn(K, N) * 0.05, torch.randn(M, N)
    a[128:] *= 2.0 ** -23
    with ops.f32_matmul("split3"):
        c = torch.empty(M, N, device=DEV)
        rd = torch.empty(M // S, N // 64, S, device=DEV)
        ops.gemm(dev(a), K, 1, dev(w2), 1, N, c, N, M, N, K, rowdot=(dev(other), rd, S))
        c2 = c.clone()
        ops.gemm(dev(a), K, 1, dev(w2), 1, N, c2, N, M, N, K, accumulate=True)
    cref = a.double() @ w2.double()
    csc = a.double().abs() @ w2.double().abs()
    assert float(_row_err(c, cref, csc).max()) < FP32_BAR
    assert float(_row_err(c2, 2 * cref, 2 * csc).max()) < FP32_BAR
    rd_ref = (cref * other.double()).view(M // S, S, N // 64, 64).sum(-1).permute(0, 2, 1)
    rd_sc = (csc * other.double().abs()).view(M // S, S, N // 64, 64).sum(-1).permute(0, 2, 1)
    assert float(((rd.cpu().double() - rd_ref).abs() / rd_sc).max()) < 4 * FP32_BAR


def test_gemm_weight_gradient_layout_needs_no_guard():
    """dW = dY^T X contracts over the tokens: rows 2^-24 below the rest add terms 2^-24 of the rest - their lost relative
    precision is an absolute error far below the result's, so the error stays relative to sum |a_k b_k| at the fp32 level."""
    torch.manual_seed(5)
    T, N, K = 4096, 384, 256
    dy, x = torch.randn(T, N) * 1e-3, torch.randn(T, K)
    dy[T // 2:] *= 2.0 ** -24
    x[: T // 4] *= 2.0 ** -20
    dy[17, 5] *= 2.0 ** 20
    with ops.f32_matmul("split3"):
        dW = torch.zeros(N, K, device=DEV)
        ops.linear_wgrad(dev(dy), N, dev(x), K, dW, K, T, N, K)
    ref, sc = dy.double().T @ x.double(), dy.double().abs().T @ x.double().abs()
    e = (dW.cpu().double() - ref).abs() / sc
    _record("split3_wgrad_rows", float(e.max()))
    assert float(e.max()) < FP32_BAR


def test_gemm_weight_operand_is_outside_the_guard():
    """Documented limit: the B operand (a parameter matrix) is not range-checked.  A weight with one element 2^30 above the rest
    costs every OTHER output column 2^(30 - 38) - shown here so that the limit is a measured number; parameters initialised
    at 0.02 and decayed by AdamW stay within a few powers of two."""
    torch.manual_seed(6)
    M, N, K = 256, 256, 256
    x, w = torch.randn(M, K), torch.randn(N, K) * 0.05
    w[9, 3] *= 2.0 ** 30
    with ops.f32_matmul("split3"):
        y = torch.empty(M, N, device=DEV)
        ops.linear_fwd(dev(x), K, dev(w), K, y, N, M, N, K)
    e = (y.cpu().double() - x.double() @ w.double().T).abs() / (x.double().abs() @ w.double().abs().T)
    cols = torch.ones(N, dtype=torch.bool); cols[9] = False
    _record("split3_weight_outlier_2^30", dict(other_columns=float(e[:, cols].max()), its_column=float(e[:, 9].max())))
    assert float(e[:, 9].max()) < FP32_BAR                  # the outlier's own column is fine
    assert 1e-5 < float(e[:, cols].max()) < 2.0 ** -6       # the others: 2^-8 per piece, as predicted


# ---- attention (the three-product fp16 form, gamer_attn_split_amax) ------------------------------------------------------
def _attn_case(B, n_items, nq, nkv, seed, qk_gain=1.0, cross=False):
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=seed, pad_rows={0: max(1, n_items // 3)})
    S = batch["input_ids"].shape[1]
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, S, nq, 64, generator=g) * qk_gain
    k = torch.randn(B, S, nkv, 64, generator=g) * qk_gain
    v = torch.randn(B, S, nkv, 64, generator=g)
    d_o = torch.randn(B, S, nq, 64, generator=g)
    self_ok, cross_ok = orc.mask_predicates(batch["attention_mask"], batch["actions"])
    return batch, S, q, k, v, d_o, (cross_ok if cross else self_ok)


def _attn_errors(batch, cross, B, S, nq, nkv, q, k, v, d_o, ok, form):
    leaves = [t.double().requires_grad_(True) for t in (q, k, v)]
    o_ref, _, _ = _attn_ref(*leaves, ok, nq, nkv, 0.125)
    (o_ref * d_o.double()).sum().backward()
    res = _run_attn(batch, cross, B, S, nq, nkv, q=q, k=k, v=v, d_o=d_o, spill=form)
    T = B * S
    refs = dict(o=o_ref.reshape(T, -1).detach(), dq=leaves[0].grad.reshape(T, -1), dk=leaves[1].grad.reshape(T, -1),
                dv=leaves[2].grad.reshape(T, -1))
    return res, refs


@pytest.mark.parametrize("cross", [False, True])
def test_attention_peaked_softmax(cross):
    """|scores| up to ~80-100 (q, k five times the unit scale): the probabilities are one-hot to fp32 precision on most rows
    and the score products carry the largest absolute error the form can produce.  Bar: the three-product form stays within
    4x of the fp32-MFMA kernels' own error against fp64 (both are dominated by the score's absolute rounding), and inside
    the grid's bars."""
    B, n_items, nq, nkv = 3, 40, 6, 3
    batch, S, q, k, v, d_o, ok = _attn_case(B, n_items, nq, nkv, 31, qk_gain=5.0, cross=cross)
    smax = float((torch.einsum("bind,bjnd->bnij", q, k.repeat_interleave(2, 2)) * 0.125).abs().max())
    assert smax > 70
    errs = {}
    for form in (False, "split", "split_h2"):
        res, refs = _attn_errors(batch, cross, B, S, nq, nkv, q, k, v, d_o, ok, form)
        errs[str(form)] = {n: float((res[n].cpu().double() - refs[n]).abs().max() / refs[n].abs().max()) for n in refs}
    _record(f"attn_peaked_cross{int(cross)}", dict(smax=smax, **errs))
    for n in ("o", "dq", "dk", "dv"):
        assert errs["split_h2"][n] < max(4 * errs["False"][n], 2e-5 if n == "o" else 5e-5), (n, errs)


@pytest.mark.parametrize("cross", [False, True])
def test_attention_backward_rows_of_dO_over_44_powers_of_two(cross):
    """dO with one outlier row (x 2^20), a block of low-gradient rows (x 2^-24) and the rest at 1: dQ is checked PER ROW against
    its own magnitude (the dQ kernel scales dO and dS per query row), dK / dV against theirs."""
    B, n_items, nq, nkv = 2, 30, 6, 3
    batch, S, q, k, v, d_o, ok = _attn_case(B, n_items, nq, nkv, 41, cross=cross)
    d_o[:, S // 2:] *= 2.0 ** -24
    d_o[0, 3] *= 2.0 ** 20
    res, refs = _attn_errors(batch, cross, B, S, nq, nkv, q, k, v, d_o, ok, "split_h2")
    T = B * S
    dq, dq_ref = res["dq"].cpu().double().view(T, nq, 64), refs["dq"].view(T, nq, 64)
    row_ref = dq_ref.abs().amax(-1)
    row_err = (dq - dq_ref).abs().amax(-1) / row_ref.clamp_min(1e-300)
    live = row_ref > 0
    e_dk = float((res["dk"].cpu().double() - refs["dk"]).abs().max() / refs["dk"].abs().max())
    e_dv = float((res["dv"].cpu().double() - refs["dv"]).abs().max() / refs["dv"].abs().max())
    _record(f"attn_dO_rows_cross{int(cross)}", dict(dq_per_row=float(row_err[live].max()), dk=e_dk, dv=e_dv))
    assert float(row_err[live].max()) < 5e-5, float(row_err[live].max())
    assert e_dk < 5e-5 and e_dv < 5e-5


def test_attention_h2_rejects_dropout_it_cannot_scale():
    """p_drop >= 0.75 would push P / (1 - p) past the fixed 2^13 scale of the three-product form: rejected loudly; the engine
    falls back to the six-product form for such a configuration."""
    B, n_items, nq, nkv = 1, 8, 2, 1
    batch, S, q, k, v, d_o, ok = _attn_case(B, n_items, nq, nkv, 3)
    with pytest.raises(RuntimeError, match="p_drop < 0.75"):
        _run_attn(batch, False, B, S, nq, nkv, p_drop=0.8, q=q, k=k, v=v, spill="split_h2")
    # the rejected call left nothing armed: the next (six-product) call is not turned into the three-product form
    res = _run_attn(batch, False, B, S, nq, nkv, p_drop=0.8, q=q, k=k, v=v, spill="split")
    assert torch.isfinite(res["o"]).all()
