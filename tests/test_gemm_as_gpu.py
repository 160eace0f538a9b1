"""Activation-stationary Linear forward (csrc/gemm_as.hip) - the kernel that takes plain nn.Linear forwards with K <= 256 and packed weight
pieces in the default product form (q|k|v, output gate, tied head: ref:SeqRec/models/generative/Qwen3Multi/model.py:93-99, 145-149, 1001).
It only engages at >= 16 k rows in production; GAMER_GEMM_AS_MIN_M = 1 brings it down to test sizes.  Reference: fp64 products, at the
bars of the tile kernel's tests (error relative to sum |a_k| |w_k|); against the tile kernel on the same inputs; exact cases."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import ops  # noqa: E402

DEV = "cuda"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


_env = ops.env_switches          # (sets the switches and has the library re-read them: they are cached per process)


def _weights(N, K, ldw=None, scale=0.05, seed=0):
    """A weight matrix inside a flat 'parameter' buffer registered as stable, with its piece planes: what Engine does once per pass."""
    g = torch.Generator().manual_seed(seed)
    ldw = ldw or K
    flat = (torch.randn(N * ldw + 8, generator=g) * scale).to(DEV)
    W = flat[:N * ldw].view(N, ldw)[:, :K]
    cache = ops.amax_reuse()
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    return flat, W, ldw, cache


def _forward(x, W, ldw, cache, M, N, K, ldy=None, as_on=True, passes=2, **kw):
    """Two passes inside the cache: the planes exist from the second one on (the first runs the in-kernel cut of the tile kernel)."""
    ldy = ldy or N
    y = None
    with _env(GAMER_GEMM_AS=int(as_on), GAMER_GEMM_AS_MIN_M=1), ops.f32_matmul("split3"), cache:
        for _ in range(passes):
            cache.reset()
            y = torch.full((M, ldy), float("nan"), device=DEV)
            ops.linear_fwd(x, x.stride(0), W, ldw, y, ldy, M, N, K, **kw)
        assert len(cache._plane_keys) > 0
    torch.cuda.synchronize()
    return y


def _err(y, x, W, N):
    xd, wd = x.double().cpu(), W.double().cpu()
    ref, sc = xd @ wd.T, xd.abs() @ wd.abs().T
    e = (y[:, :N].double().cpu() - ref).abs() / sc.clamp_min(1e-300)
    return float(e.max()), float(e.pow(2).mean().sqrt())


@pytest.mark.parametrize("K", [64, 128, 192, 256])
@pytest.mark.parametrize("M,N,ldy", [(1000, 768, 768), (517, 1041, 1056), (32, 32, 32), (4099, 100, 128), (1, 256, 256)])
def test_gemm_as_against_fp64_and_the_tile_kernel(M, N, ldy, K):
    """Ragged row counts (a last workgroup with idle waves, a last wave with idle lanes), N with a partial last slab (1041 = 32 x 32 + 17,
    100), a padded output row stride; every K the kernel instantiates."""
    flat, W, ldw, cache = _weights(N, K, seed=N + K)
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g))).to(DEV)
    y_as = _forward(x, W, ldw, cache, M, N, K, ldy=ldy, as_on=True)
    y_tile = _forward(x, W, ldw, cache, M, N, K, ldy=ldy, as_on=False)
    e_as, e_tile = _err(y_as, x, W, N), _err(y_tile, x, W, N)
    assert e_as[0] < 1e-6 and e_as[1] < 6e-8, (e_as, e_tile)
    assert e_as[0] < 1.5 * e_tile[0] + 1e-8 and e_as[1] < 1.25 * e_tile[1]
    if ldy > N:                                    # the padding columns are not written
        assert torch.isnan(y_as[:, N:]).all()
    assert torch.isfinite(y_as[:, :N]).all()


def _as_launches():
    import ctypes
    from gamer_amd import _lib
    fn = _lib.load().gamer_debug_gemm_as_launches
    fn.restype = ctypes.c_longlong
    return int(fn())


def test_gemm_as_is_really_the_kernel_that_ran():
    """On well-scaled data the two kernels return the same BITS (a power-of-two scale - per row here, per tensor there - does not change
    an fp16 rounding, and both add the piece products in the same order), so the results cannot tell which ran: the library counts the
    launches.  First pass of a cache = no planes yet = tile kernel; GAMER_GEMM_AS=0 and rows below the bar = tile kernel."""
    M, N, K = 2048, 768, 256
    flat, W, ldw, cache = _weights(N, K)
    x = torch.randn(M, K, device=DEV) * 3
    n0 = _as_launches()
    y_as = _forward(x, W, ldw, cache, M, N, K, as_on=True)
    assert _as_launches() == n0 + 1
    y_tile = _forward(x, W, ldw, cache, M, N, K, as_on=False)
    assert _as_launches() == n0 + 1
    assert float((y_as - y_tile).abs().max()) < 2e-6 * float(y_tile.abs().max())
    assert torch.equal(y_as, _forward(x, W, ldw, cache, M, N, K, as_on=True))          # repeats itself bit for bit
    assert _as_launches() == n0 + 3                  # (the cache knows the weights by now: both passes had planes)
    with _env(GAMER_GEMM_AS=1), ops.f32_matmul("split3"), cache:                        # production bar: 16 k rows
        for _ in range(2):
            cache.reset()
            ops.linear_fwd(x, K, W, ldw, torch.empty(M, N, device=DEV), N, M, N, K)
    assert _as_launches() == n0 + 3


def test_gemm_as_rows_of_any_magnitude_keep_their_relative_error():
    """A per-row power-of-two scale: rows 2^-40 .. 2^+40 apart in one launch come out at the same relative error, with no range guard
    (the tile kernel redoes such tiles on the fp32 MFMA); small integers exactly; zero rows give zeros; an Inf poisons its row only."""
    M, N, K = 640, 384, 256
    flat, W, ldw, cache = _weights(N, K, scale=0.3)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(M, K, generator=g)
    mag = torch.exp2(torch.randint(-40, 41, (M, 1), generator=g).float())
    x = x * mag
    x[5] = 0.0
    y = _forward(x.to(DEV), W, ldw, cache, M, N, K)
    xd, wd = x.double(), W.double().cpu()
    e = (y.double().cpu() - xd @ wd.T).abs() / (xd.abs() @ wd.abs().T).clamp_min(1e-300)
    assert float(e.max()) < 1e-6 and float(e.pow(2).mean().sqrt()) < 6e-8
    assert torch.equal(y[5].cpu(), torch.zeros(N))
    # integers
    flat2 = torch.zeros(N * K + 8, device=DEV)
    wi = torch.randint(-30, 31, (N, K)).float()
    flat2[:N * K] = wi.flatten().to(DEV)
    cache2 = ops.amax_reuse()
    cache2.stable_range(flat2.data_ptr(), flat2.numel() * 4)
    cache2.planes = torch.zeros(flat2.numel(), dtype=torch.float32, device=DEV)
    xi = torch.randint(-30, 31, (M, K)).float()
    yi = _forward(xi.to(DEV), flat2[:N * K].view(N, K), K, cache2, M, N, K)
    assert torch.equal(yi.cpu().double(), xi.double() @ wi.double().T)
    xinf = xi.clone()
    xinf[7, 3] = float("inf")
    yinf = _forward(xinf.to(DEV), flat2[:N * K].view(N, K), K, cache2, M, N, K)
    assert not torch.isfinite(yinf[7].cpu()).all()
    ok = torch.ones(M, dtype=torch.bool)
    ok[7] = False
    assert torch.equal(yinf.cpu()[ok].double(), (xi.double() @ wi.double().T)[ok])


@pytest.mark.parametrize("M", [640, 300])
def test_gemm_as_reports_the_maximum_of_the_columns_it_is_asked_for(M):
    """gamer_gemm_desc.amax_c: the v columns of a q|k|v projection (what the attention kernels cut V with)."""
    from test_ops_gpu import _slot_value
    H, nq, nkv = 256, 6, 3
    QKV = (nq + 2 * nkv) * 64
    g = torch.Generator().manual_seed(1)
    flat = (torch.randn(QKV * H + 8, generator=g) * 0.3).to(DEV)
    W = flat[:QKV * H].view(QKV, H)
    x = torch.randn(M, H, device=DEV)
    col0 = (nq + nkv) * 64
    cache = ops.amax_reuse(everything=True)
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    with _env(GAMER_GEMM_AS=1, GAMER_GEMM_AS_MIN_M=1), ops.f32_matmul("split3"), cache:
        for _ in range(2):
            cache.reset()
            y = torch.zeros(M, QKV, device=DEV)
            v = y[:, col0:]
            ops.linear_fwd(x, H, W, H, y, QKV, M, QKV, H, c_amax=(v, col0))
        assert len(cache._plane_keys) > 0
        key = cache._key(v.data_ptr(), (1, 0, M, nkv * 64, QKV))
        assert key in cache.pending, "the producer did not open a slot"
        torch.cuda.synchronize()
        assert _slot_value(cache.pending[key]) == float(v.abs().max())
    ref = x.double().cpu() @ W.double().cpu().T
    assert float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 1e-6


@pytest.mark.parametrize("K", [256, 320])
def test_gemm_as_grouped_experts(K):
    """The experts' gate|up projection: rows sorted by expert, one weight matrix per segment (an empty segment, one shorter than a
    wave, boundaries that are no multiple of anything); K = 320 = the injecting layers' input width, which the kernel leaves to the tile kernel."""
    T, N, E = 3000, 1024, 6
    offs = torch.tensor([0, 700, 700, 717, 1500, 2100, T], dtype=torch.int32, device=DEV)
    g = torch.Generator().manual_seed(K)
    flat = (torch.randn(E * N * K + 8, generator=g) * 0.05).to(DEV)
    W = flat[:E * N * K].view(E * N, K)
    x = (torch.randn(T, K, generator=g) * torch.exp(torch.randn(T, 1, generator=g))).to(DEV)
    cache = ops.amax_reuse()
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    out = {}
    for as_on in (1, 0):
        n0 = _as_launches()
        with _env(GAMER_GEMM_AS=as_on, GAMER_GEMM_AS_MIN_M=1), ops.f32_matmul("split3"), cache:
            for _ in range(2):
                cache.reset()
                y = torch.full((T, N), float("nan"), device=DEV)
                ops.linear_fwd(x, K, W, K, y, N, T, N, K, strideB=N * K, groups=E, group_offsets=offs)
        torch.cuda.synchronize()
        assert (_as_launches() > n0) == bool(as_on and K <= 256)          # (K = 320 stays on the tile kernel)
        out[as_on] = y
    o = offs.cpu().tolist()
    for e in range(E):
        xe, we = x[o[e]:o[e + 1]].double().cpu(), W[e * N:(e + 1) * N].double().cpu()
        ref, sc = xe @ we.T, xe.abs() @ we.abs().T
        err = (out[1][o[e]:o[e + 1]].double().cpu() - ref).abs() / sc.clamp_min(1e-300)
        if err.numel():
            assert float(err.max()) < 1e-6, e
    assert float((out[1] - out[0]).abs().max()) < 2e-6 * float(out[0].abs().max())


# ---- W row-contiguous: input gradients of the layers with 256 output features (o_proj, the cross block's gate, the experts' down projection)
def _param_cache(flat):
    cache = ops.amax_reuse(everything=True)
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    return cache


def _both(cache, fn):
    """fn() under the tile kernel and under the activation-stationary one (second pass of the cache: planes exist); returns their outputs
    and checks by the launch counter that the second really ran it."""
    outs = {}
    for as_on in (0, 1):
        n0 = _as_launches()
        with _env(GAMER_GEMM_AS=as_on, GAMER_GEMM_AS_MIN_M=1, GAMER_GEMM_AS_RC=2, GAMER_GEMM_OS=0), ops.f32_matmul("split3"), cache:
            for _ in range(2):
                cache.reset()
                outs[as_on] = fn()
            assert len(cache._plane_keys) > 0
        torch.cuda.synchronize()
        assert (_as_launches() > n0) == bool(as_on)
    return outs[0], outs[1]


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("M,N", [(1000, 384), (517, 512), (33, 100), (4099, 256)])
def test_gemm_as_input_gradient_layout(M, N, accumulate):
    K = 256
    g = torch.Generator().manual_seed(M + N)
    flat = (torch.randn(K * N + 8, generator=g) * 0.05).to(DEV)
    W = flat[:K * N].view(K, N)
    dy = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g)) * 1e-3).to(DEV)
    base = torch.randn(M, N, generator=g).to(DEV) * 1e-3

    def run():
        dx = base.clone() if accumulate else torch.full((M, N), float("nan"), device=DEV)
        ops.linear_dgrad(dy, K, W, N, dx, N, M, K, N, accumulate=accumulate)
        return dx
    tile, got = _both(_param_cache(flat), run)
    ref = dy.double().cpu() @ W.double().cpu() + (base.double().cpu() if accumulate else 0)
    sc = dy.double().cpu().abs() @ W.double().cpu().abs() + (base.double().cpu().abs() if accumulate else 0)
    e = (got.double().cpu() - ref).abs() / sc.clamp_min(1e-300)
    assert float(e.max()) < 1e-6 and float(e.pow(2).mean().sqrt()) < 7e-8
    assert float((got - tile).abs().max()) <= 2e-6 * float(tile.abs().max())


def test_gemm_as_row_dot_epilogue():
    """o_proj input gradient: C as without the epilogue and out[b][head][i] = C[m, head] . other[m, head] (the attention backward's delta)."""
    B, S, heads, K = 20, 64, 6, 256
    M, N = B * S, heads * 64
    g = torch.Generator().manual_seed(3)
    flat = (torch.randn(K * N + 8, generator=g) * 0.05).to(DEV)
    W = flat[:K * N].view(K, N)
    dy, other = (torch.randn(M, K, generator=g) * 1e-2).to(DEV), torch.randn(M, N, generator=g).to(DEV)

    def run():
        dx = torch.full((M, N), float("nan"), device=DEV)
        out = torch.full((B, heads, S), float("nan"), device=DEV)
        ops.linear_dgrad(dy, K, W, N, dx, N, M, K, N, rowdot=(other, out, S), c_amax=(dx, 0))
        return dx, out
    (dx0, out0), (dx1, out1) = _both(_param_cache(flat), run)
    ref = dy.double().cpu() @ W.double().cpu()
    assert float((dx1.double().cpu() - ref).abs().max()) < 2e-6 * float(ref.abs().max())
    want = (ref.view(B, S, heads, 64) * other.double().cpu().view(B, S, heads, 64)).sum(-1).permute(0, 2, 1)
    assert float((out1.double().cpu() - want).abs().max()) < 5e-6 * float(want.abs().max())
    assert float((out1 - out0).abs().max()) < 5e-6 * float(out0.abs().max())
    assert float((dx1 - dx0).abs().max()) <= 2e-6 * float(dx0.abs().max())


@pytest.mark.parametrize("T,offsets", [(1024, None), (700, [0, 0, 130, 131, 389, 389, 700]), (37, None), (900, [64, 64, 200, 333, 600, 600, 850])])
def test_gemm_as_swiglu_backward_epilogue(T, offsets):
    """The experts' down-projection input gradient with the SwiGLU backward in its epilogue (d(hm) never stored), plain and grouped
    (ragged segments, empty experts, rows outside every segment), dropout on: the same d gate | d up as the tile kernel's epilogue (same
    dropout mask function, same formula) and the same maximum in the slot."""
    from test_ops_gpu import _slot_value
    H, I, p_drop, seed = 256, 512, 0.2, 13
    E = 6 if offsets is not None else 1
    g = torch.Generator().manual_seed(T)
    flat = (torch.randn(E * H * I + 8, generator=g) * 0.05).to(DEV)
    Wd = flat[:E * H * I].view(E * H, I)
    t0 = (torch.randn(T, H, generator=g) * 0.1).to(DEV)
    gu0 = torch.randn(T, 2 * I, generator=g)
    grp = {}
    if offsets is not None:
        grp = dict(groups=E, group_offsets=torch.tensor(offsets, dtype=torch.int32, device=DEV), strideB=H * I)
    cache = _param_cache(flat)
    slots = {}

    def run():
        got = gu0.to(DEV)
        dummy = torch.full((T, I), 5.0, device=DEV)
        ops.gemm(t0, H, 1, Wd, 1, I, dummy, I, T, I, H, p_drop=p_drop, seed=seed, swiglu_bwd=(got, 2 * I), **grp)
        key = cache._key(got.data_ptr(), (1, 0, 1, T * 2 * I, T * 2 * I))
        assert key in cache.pending
        torch.cuda.synchronize()
        slots[len(slots)] = (_slot_value(cache.pending[key]), float(got.abs().max()))
        assert bool((dummy == 5.0).all()), "C must not be written"
        return got
    tile, got = _both(cache, run)
    scale = float(tile.abs().max())
    assert float((got - tile).abs().max()) <= 2e-6 * scale, float((got - tile).abs().max()) / scale
    for k_, (slot, mx) in slots.items():
        if offsets is None or offsets[0] == 0 and offsets[-1] == T:
            assert slot == mx, k_
        else:
            assert slot <= mx            # (rows outside every segment keep their values and are not part of the maximum)
    if offsets is not None and offsets[0] > 0:
        assert torch.equal(got[:offsets[0]].cpu(), gu0[:offsets[0]])
        assert torch.equal(got[offsets[-1]:].cpu(), gu0[offsets[-1]:])


# ---- the SwiGLU forward as the epilogue of the experts' gate|up projection (gamer_gemm_desc.sw_hm) ---------------------------------
@pytest.mark.parametrize("p_drop", [0.0, 0.2])
@pytest.mark.parametrize("table", [False, True])
def test_gemm_as_swiglu_forward_epilogue_equals_the_two_launches(table, p_drop):
    """One call leaves gate|up AND hm = dropout(silu(gate + tg) * (up + tu)): the bits of the projection followed by
    gamer_swiglu_fwd_ld(_tbl) - the epilogue runs the same arithmetic on the values it stores - with the rows grouped by (expert,
    behaviour) (24 groups, four per weight matrix, an empty group, ragged boundaries) when the row table is given; the maximum of hm
    reaches the slot its consumer will take.  Without packed pieces (first pass) / with GAMER_GEMM_AS_SWIGLU=0 the library runs
    the two launches itself: same results."""
    from test_ops_gpu import _slot_value
    T, I, K, E, NB1 = 3000, 512, 256, 6, 4
    N = 2 * I
    g = torch.Generator().manual_seed(11 + int(table))
    flat = (torch.randn(E * N * K + 8, generator=g) * 0.05).to(DEV)
    W = flat[:E * N * K].view(E * N, K)
    x = (torch.randn(T, K, generator=g) * torch.exp(torch.randn(T, 1, generator=g))).to(DEV)
    G = E * NB1 if table else E
    cuts = sorted(torch.randint(0, T + 1, (G - 1,), generator=g).tolist())
    cuts[3] = cuts[2]                                               # an empty group
    offs = torch.tensor([0] + cuts + [T], dtype=torch.int32, device=DEV)
    tbl = torch.randn(G, N, generator=g).to(DEV) if table else None
    o = offs.cpu().tolist()
    row_group = torch.cat([torch.full((o[i + 1] - o[i],), i, dtype=torch.int32) for i in range(G)]).to(DEV) if table else None
    grp = dict(strideB=N * K, groups=G, group_offsets=offs, group_div=NB1 if table else 0)
    cache = ops.amax_reuse()
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)

    def run(fused_call, swiglu_kernel_on):
        gu = hm = None
        slot = None
        with _env(GAMER_GEMM_AS=1, GAMER_GEMM_AS_MIN_M=1, GAMER_GEMM_AS_SWIGLU=int(swiglu_kernel_on)), ops.f32_matmul("split3"), cache:
            for _ in range(2):
                cache.reset()
                gu = torch.full((T, N), float("nan"), device=DEV)
                hm = torch.full((T, I), float("nan"), device=DEV)
                if fused_call:
                    ops.gemm(x, K, 1, W, K, 1, gu, N, T, N, K, p_drop=p_drop, seed=77, swiglu_fwd=(hm, tbl, row_group), **grp)
                else:
                    ops.gemm(x, K, 1, W, K, 1, gu, N, T, N, K, **grp)
                    if table:
                        ops.swiglu_fwd_ld_tbl(gu, N, T, I, p_drop, 77, hm, tbl, row_group)
                    else:
                        ops.swiglu_fwd_ld(gu, N, T, I, p_drop, 77, hm)
            key = cache._key(hm.data_ptr(), (1, 0, 1, T * I, T * I))
            slot = cache.pending.get(key)
            torch.cuda.synchronize()
            smax = _slot_value(slot) if slot is not None else None
        return gu, hm, smax
    n0 = _as_launches()
    gu_f, hm_f, smax_f = run(True, True)
    assert _as_launches() > n0
    gu_r, hm_r, _ = run(False, True)
    if table:
        # (rows in groups that share a weight matrix are a form of the 128 x 128 kernel when the projection runs alone: its gate|up agree to
        # rounding; the epilogue's hm must be the bits of the stand-alone kernel on the gate|up values the SAME call stored)
        assert float((gu_f - gu_r).abs().max()) < 2e-6 * float(gu_r.abs().max())
        hm_own = torch.empty_like(hm_f)
        ops.swiglu_fwd_ld_tbl(gu_f, N, T, I, p_drop, 77, hm_own, tbl, row_group)
        assert torch.equal(hm_f, hm_own)
    else:
        assert torch.equal(gu_f, gu_r) and torch.equal(hm_f, hm_r)
    assert smax_f is not None and smax_f == float(hm_f.abs().max())
    gu_b, hm_b, smax_b = run(True, False)                           # the library's own two launches behind the same call
    assert torch.equal(gu_b, gu_r) and torch.equal(hm_b, hm_r) and smax_b == float(hm_b.abs().max())
    # and against fp64
    wsel = torch.cat([torch.full((o[i + 1] - o[i],), (i // NB1) if table else i, dtype=torch.long) for i in range(G)])
    ref = torch.einsum("tk,tnk->tn", x.double().cpu(), W.double().cpu().view(E, N, K)[wsel])
    assert float((gu_f.double().cpu() - ref).abs().max()) < 2e-6 * float(ref.abs().max())
    if p_drop == 0.0:
        full = ref + (tbl.double().cpu()[row_group.cpu().long()] if table else 0.0)
        want = torch.nn.functional.silu(full[:, :I]) * full[:, I:]
        assert float((hm_f.double().cpu() - want).abs().max()) < 5e-6 * float(want.abs().max())
