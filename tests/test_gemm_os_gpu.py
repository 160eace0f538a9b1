"""Output-stationary input gradient (csrc/gemm_os.hip): dX = dY W for nn.Linear layers with 256 input features (q|k|v projections, tied
head: ref:SeqRec/models/generative/Qwen3Multi/model.py:93-99, 1001), three-product form.  It engages at >= 16 k rows in production;
GAMER_GEMM_OS_MIN_M = 1 brings it down to test sizes.  Reference: fp64, at the bars of the tile kernel's tests (error relative to
sum |dy_k| |w_k|), and the tile kernel on the same inputs; the library counts the launches."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import _lib, ops  # noqa: E402

DEV = "cuda"


_env = ops.env_switches          # (sets the switches and has the library re-read them: they are cached per process)


def _launches():
    fn = _lib.load().gamer_debug_gemm_os_launches
    fn.restype = ctypes.c_longlong
    return int(fn())


def _weights(N_out, H, scale=0.05, seed=0):
    g = torch.Generator().manual_seed(seed)
    flat = (torch.randn(N_out * H + 8, generator=g) * scale).to(DEV)
    cache = ops.amax_reuse()
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    return flat, flat[:N_out * H].view(N_out, H), cache


def _dgrad(dy, ldy, W, cache, M, N_out, H, os_on, **kw):
    with _env(GAMER_GEMM_OS=int(os_on), GAMER_GEMM_OS_MIN_M=1), ops.f32_matmul("split3"), cache:
        for _ in range(2):                       # (the planes exist from the second pass of a cache on)
            cache.reset()
            dx = torch.full((M, H), float("nan"), device=DEV)
            ops.linear_dgrad(dy, ldy, W, H, dx, H, M, N_out, H, **kw)
        assert len(cache._plane_keys) > 0
    torch.cuda.synchronize()
    return dx


def _err(dx, dy, W, N_out):
    dyd, wd = dy[:, :N_out].double().cpu(), W.double().cpu()
    e = (dx.double().cpu() - dyd @ wd).abs() / (dyd.abs() @ wd.abs()).clamp_min(1e-300)
    return float(e.max()), float(e.pow(2).mean().sqrt())


@pytest.mark.parametrize("M", [1000, 256, 33, 4099])
@pytest.mark.parametrize("N_out,ldy", [(768, 768), (1041, 1056), (1024, 1024), (20, 32), (4, 4)])
def test_gemm_os_against_fp64_and_the_tile_kernel(N_out, ldy, M):
    """Contractions that are no multiple of the 32-wide block (1041 = 32 x 32 + 17, 20, 4) with a padded row stride whose padding
    holds NaN (it must never be read into a product), ragged row counts (idle waves, idle lanes)."""
    H = 256
    flat, W, cache = _weights(N_out, H, seed=N_out)
    g = torch.Generator().manual_seed(M + N_out)
    dy = torch.full((M, ldy), float("nan"))
    dy[:, :N_out] = torch.randn(M, N_out, generator=g) * torch.exp(torch.randn(M, N_out, generator=g)) * 1e-3
    dy = dy.to(DEV)
    n0 = _launches()
    got = _dgrad(dy, ldy, W, cache, M, N_out, H, True)
    assert _launches() > n0
    n1 = _launches()
    ref = _dgrad(dy, ldy, W, cache, M, N_out, H, False)
    assert _launches() == n1
    e_os, e_tile = _err(got, dy, W, N_out), _err(ref, dy, W, N_out)
    assert torch.isfinite(got).all()
    assert e_os[0] < 1e-6 and e_os[1] < 7e-8, (e_os, e_tile)          # (rms: 2^-24 = 6e-8 is the rounding of the fp32 result itself)
    assert e_os[0] < 1.5 * e_tile[0] + 1e-8 and e_os[1] < 1.25 * e_tile[1]


@pytest.mark.parametrize("guard", [1, 0])
def test_gemm_os_rows_far_below_the_tensor_maximum(guard):
    """The row-range guard: rows 2^-30 below the tensor's largest magnitude (a loss spike elsewhere in the batch) keep their relative
    precision - the workgroup repeats its contraction with a scale per row; with the guard off (gamer_split3_guard(0)) they do not,
    which is what tells that the guard is what did it."""
    M, N_out, H = 1024, 768, 256
    flat, W, cache = _weights(N_out, H, scale=0.3)
    g = torch.Generator().manual_seed(7)
    dy = torch.randn(M, N_out, generator=g)
    dy[100:400] *= 2.0 ** -30
    dy[700] = 0.0
    dy = dy.to(DEV)
    lib = _lib.load()
    lib.gamer_split3_guard(guard)
    try:
        got = _dgrad(dy, N_out, W, cache, M, N_out, H, True)
    finally:
        lib.gamer_split3_guard(1)
    dyd, wd = dy.double().cpu(), W.double().cpu()
    e = (got.double().cpu() - dyd @ wd).abs() / (dyd.abs() @ wd.abs()).clamp_min(1e-300)
    small = float(e[100:400].max())
    assert float(e[:100].max()) < 1e-6 and float(e[400:].max()) < 1e-6
    assert torch.equal(got[700].cpu(), torch.zeros(H))
    if guard:
        assert small < 1e-6
    else:
        assert small > 1e-5


def test_gemm_os_integers_and_the_default_bar():
    M, N_out, H = 512, 192, 256
    flat = torch.zeros(N_out * H + 8, device=DEV)
    wi = torch.randint(-30, 31, (N_out, H)).float()
    flat[:N_out * H] = wi.flatten().to(DEV)
    cache = ops.amax_reuse()
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    dyi = torch.randint(-30, 31, (M, N_out)).float()
    got = _dgrad(dyi.to(DEV), N_out, flat[:N_out * H].view(N_out, H), cache, M, N_out, H, True)
    assert torch.equal(got.cpu().double(), dyi.double() @ wi.double())
    n0 = _launches()
    with _env(GAMER_GEMM_OS=1), ops.f32_matmul("split3"), cache:          # production bar: 16 k rows
        for _ in range(2):
            cache.reset()
            ops.linear_dgrad(dyi.to(DEV), N_out, flat[:N_out * H].view(N_out, H), H, torch.empty(M, H, device=DEV), H, M, N_out, H)
    assert _launches() == n0


def test_gemm_os_grouped_experts():
    """The experts' gate|up input gradient at d_in = 256: rows sorted by expert, one W per segment (an empty segment, one shorter than a
    wave, boundaries that are no multiple of anything)."""
    T, N_out, H, E = 3000, 1024, 256, 6
    offs = torch.tensor([0, 700, 700, 717, 1500, 2100, T], dtype=torch.int32, device=DEV)
    g = torch.Generator().manual_seed(5)
    flat = (torch.randn(E * N_out * H + 8, generator=g) * 0.05).to(DEV)
    W = flat[:E * N_out * H].view(E * N_out, H)
    dy = (torch.randn(T, N_out, generator=g) * 1e-3).to(DEV)
    cache = ops.amax_reuse()
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    grp = dict(groups=E, group_offsets=offs, strideB=N_out * H)
    n0 = _launches()
    got = _dgrad(dy, N_out, W, cache, T, N_out, H, True, **grp)
    assert _launches() > n0
    ref = _dgrad(dy, N_out, W, cache, T, N_out, H, False, **grp)
    o = offs.cpu().tolist()
    for e in range(E):
        want = dy[o[e]:o[e + 1]].double().cpu() @ W[e * N_out:(e + 1) * N_out].double().cpu()
        if want.numel():
            assert float((got[o[e]:o[e + 1]].double().cpu() - want).abs().max()) < 2e-6 * float(want.abs().max()), e
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


# ---- the Linear FORWARD of layers with 256 output features on W's transposed pieces (gamer_gemm_desc.b_planes_t) -------------------
def _fwd_case(M, K, E, resid, p_drop, osf, seed):
    """y = (resid +) dropout(x W^T) through ops.gemm in the forward layout; osf: the output-stationary kernel on / off."""
    H = 256
    g = torch.Generator().manual_seed(seed)
    flat = (torch.randn(E * H * K + 8, generator=g) * 0.05).to(DEV)
    W = flat[:E * H * K].view(E * H, K)
    x = (torch.randn(M, K + 4, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(DEV)[:, :K]      # (padded row stride)
    cache = ops.amax_reuse()
    cache.stable_range(flat.data_ptr(), flat.numel() * 4)
    cache.planes = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    cache.planes_t = torch.zeros(flat.numel(), dtype=torch.float32, device=DEV)
    kw = {}
    offs = None
    if E > 1:
        cuts = sorted(torch.randint(0, M + 1, (E - 1,), generator=g).tolist())
        cuts[1] = cuts[0]                                       # an empty segment
        offs = torch.tensor([0] + cuts + [M], dtype=torch.int32, device=DEV)
        kw = dict(groups=E, group_offsets=offs, strideB=H * K)
    r = perm = None
    if resid:
        r = torch.randn(M, H, generator=g).to(DEV)
        perm = torch.randperm(M, generator=g).to(torch.int32).to(DEV) if E > 1 else None
        kw.update(resid=r, row_map=perm, p_drop=p_drop, seed=1234)
    n0 = _launches()
    with _env(GAMER_GEMM_OSF=int(osf), GAMER_GEMM_OS_MIN_M=1), ops.f32_matmul("split3"), cache:
        for _ in range(2):
            cache.reset()
            cache.register_transposed(W, E, H, K)
            y = torch.full((M, H), float("nan"), device=DEV)
            ops.gemm(x, x.stride(0), 1, W, K, 1, y, H, M, H, K, **kw)
    torch.cuda.synchronize()
    assert (_launches() > n0) == bool(osf)
    return x, W, y, r, perm, offs


@pytest.mark.parametrize("M", [1000, 4099, 37])
@pytest.mark.parametrize("K,E", [(384, 1), (512, 6), (64, 1)])
def test_gemm_os_forward_on_transposed_pieces_plain_store(K, E, M):
    x, W, y, _, _, offs = _fwd_case(M, K, E, False, 0.0, True, seed=K + M)
    _, _, y_tile, _, _, _ = _fwd_case(M, K, E, False, 0.0, False, seed=K + M)
    o = [0, M] if offs is None else offs.cpu().tolist()
    for e in range(E):
        xs = x[o[e]:o[e + 1]].double().cpu()
        if xs.numel() == 0:
            continue
        wd = W[e * 256:(e + 1) * 256].double().cpu()
        ref, sc = xs @ wd.T, xs.abs() @ wd.abs().T
        err = ((y[o[e]:o[e + 1]].double().cpu() - ref).abs() / sc.clamp_min(1e-300))
        assert float(err.max()) < 1.5e-6 and float(err.pow(2).mean().sqrt()) < 1e-7, (e, float(err.max()))
    assert float((y - y_tile).abs().max()) <= 3e-6 * float(y_tile.abs().max())


@pytest.mark.parametrize("p_drop", [0.0, 0.2])
@pytest.mark.parametrize("K,E", [(384, 1), (512, 6)])
def test_gemm_os_forward_residual_epilogue_equals_the_tile_kernels(K, E, p_drop):
    """resid + dropout(x W^T), scattered through the row map for the experts: the same dropout mask function as csrc/gemm.hip (the
    element's flat index in the OUTPUT), so the two kernels agree to the rounding of the product."""
    M = 3000
    x, W, y, r, perm, offs = _fwd_case(M, K, E, True, p_drop, True, seed=7 + K)
    _, _, y_tile, _, _, _ = _fwd_case(M, K, E, True, p_drop, False, seed=7 + K)
    assert bool(torch.isfinite(y).all())
    d = (y - y_tile).abs()
    assert float(d.max()) <= 5e-6 * float(y_tile.abs().max())
    if p_drop == 0.0:
        o = [0, M] if offs is None else offs.cpu().tolist()
        rows = torch.arange(M) if perm is None else perm.cpu().long()
        want = torch.empty(M, 256, dtype=torch.float64)
        for e in range(E):
            seg = slice(o[e], o[e + 1])
            want[rows[seg]] = r.double().cpu()[rows[seg]] + x[seg].double().cpu() @ W[e * 256:(e + 1) * 256].double().cpu().T
        assert float((y.double().cpu() - want).abs().max()) < 5e-6 * float(want.abs().max())
