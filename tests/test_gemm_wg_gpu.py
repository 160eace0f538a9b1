"""Large-tile weight-gradient kernel (csrc/gemm_wg.hip: 256 x 256 tiles of dW, four waves of 512 registers) against the 128 x 128 kernel
of csrc/gemm.hip: both sum a chunk of tokens into the same partial tiles in the same order, so dW must agree BIT FOR BIT at the same
token chunk - on whole tiles (what the kernel takes by default), on ragged shapes (GAMER_GEMM_WG=2: every shape), ragged chunks,
expert segments with arbitrary boundaries (an empty one included) - and against fp64.  The library counts the launches: the results
cannot tell which kernel ran.  Shapes of ref:SeqRec/models/generative/Qwen3Multi/model.py:93-99, 145-149 and Qwen3Moe/FFN.py:25-27."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import _lib, ops  # noqa: E402

DEV = "cuda"


_env = ops.env_switches          # (sets the switches and has the library re-read them: they are cached per process)


def _launches():
    fn = _lib.load().gamer_debug_gemm_wg_launches
    fn.restype = ctypes.c_longlong
    return int(fn())


def _wgrad(dy, ldy, x, T, N, K, kchunk, mode, **grp):
    E = grp.get("groups", 1)
    dW = torch.zeros(E * N, K, device=DEV)
    with _env(GAMER_GEMM_WG=mode), ops.f32_matmul("split3"):
        ops.linear_wgrad(dy, ldy, x, K, dW, K, T, N, K, kchunk=kchunk, **grp)
    torch.cuda.synchronize()
    return dW


@pytest.mark.parametrize("T,kchunk", [(4096, 1024), (5000, 2080), (777, 256), (33, 32)])
@pytest.mark.parametrize("N,K,ldy", [(768, 256, 768), (256, 512, 256), (256, 384, 256), (1041, 256, 1056), (1024, 320, 1024), (100, 36, 128)])
def test_gemm_wg_bits_of_the_tile_kernel_and_fp64(N, K, ldy, T, kchunk):
    g = torch.Generator().manual_seed(N + K + T)
    x = (torch.randn(T, K, generator=g) * torch.exp(torch.randn(T, K, generator=g))).to(DEV)
    dy = (torch.randn(T, ldy, generator=g) * 1e-3).to(DEV)
    n0 = _launches()
    got = _wgrad(dy, ldy, x, T, N, K, kchunk, 2)
    assert _launches() == n0 + 1
    ref = _wgrad(dy, ldy, x, T, N, K, kchunk, 0)
    assert _launches() == n0 + 1
    assert torch.equal(got, ref)
    dyd, xd = dy[:, :N].double().cpu(), x.double().cpu()
    e = (got.double().cpu() - dyd.T @ xd).abs() / (dyd.abs().T @ xd.abs()).clamp_min(1e-300)
    assert float(e.max()) < 1e-6 and float(e.pow(2).mean().sqrt()) < 6e-8
    # the default rule: whole 256 x 256 tiles only
    whole = N % 256 == 0 and K % 256 == 0
    _wgrad(dy, ldy, x, T, N, K, kchunk, 1)
    assert _launches() == n0 + 1 + int(whole)


@pytest.mark.parametrize("N,K", [(256, 512), (1024, 256), (1024, 320)])
def test_gemm_wg_expert_segments(N, K):
    """Grouped form: token segments with arbitrary boundaries (one empty, one shorter than a stage, none a multiple of the chunk)."""
    T, E = 6000, 6
    offs = torch.tensor([0, 1500, 1500, 1517, 3003, 4100, T], dtype=torch.int32, device=DEV)
    g = torch.Generator().manual_seed(N)
    x = torch.randn(T, K, generator=g).to(DEV) * 2
    dy = (torch.randn(T, N, generator=g) * 1e-2).to(DEV)
    grp = dict(groups=E, group_offsets=offs, strideC=N * K)
    for kchunk in (256, 1056):
        n0 = _launches()
        got = _wgrad(dy, N, x, T, N, K, kchunk, 2, **grp)
        assert _launches() == n0 + 1
        assert torch.equal(got, _wgrad(dy, N, x, T, N, K, kchunk, 0, **grp))
        o = offs.cpu().tolist()
        for e_ in range(E):
            ref = dy[o[e_]:o[e_ + 1]].double().cpu().T @ x[o[e_]:o[e_ + 1]].double().cpu()
            blk = got[e_ * N:(e_ + 1) * N].double().cpu()
            assert float((blk - ref).abs().max()) <= 2e-6 * max(1e-30, float(ref.abs().max())), e_


def test_gemm_wg_accumulates_into_dw_and_repeats_itself():
    T, N, K = 3000, 512, 256
    x, dy = torch.randn(T, K, device=DEV), torch.randn(T, N, device=DEV) * 1e-2
    with _env(GAMER_GEMM_WG=1), ops.f32_matmul("split3"):
        dW = torch.full((N, K), 3.0, device=DEV)
        ops.linear_wgrad(dy, N, x, K, dW, K, T, N, K, kchunk=1056)
        dW2 = torch.full((N, K), 3.0, device=DEV)
        ops.linear_wgrad(dy, N, x, K, dW2, K, T, N, K, kchunk=1056)
    assert torch.equal(dW, dW2)
    ref = 3.0 + dy.double().T @ x.double()
    assert float((dW.double() - ref).abs().max()) < 2e-6 * float(ref.abs().max())
