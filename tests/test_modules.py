"""Discriminative baselines' encoder: oracle and HIP modules against
tests/golden/modules_small.npz (outputs and gradients of the REAL ``SeqRec.modules.layers.transformer`` classes,
oracle/make_golden_modules.py).  Tolerances: fp32 against fp32 reference, 2e-5 relative to the tensor's abs-max
for outputs, 2e-4 for gradients (sums over B*S rows)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import modules_oracle as mo

FX = os.path.join(os.path.dirname(__file__), "golden", "modules_small.npz")


def _load(name):
    z = np.load(FX)
    meta = json.loads(str(z["meta_json"]))["cases"][name]
    pre = name + "/sd/"
    sd = {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    t = lambda k: torch.from_numpy(z[f"{name}/{k}"])
    return z, meta, sd, t("x"), t("mask"), t("out"), t("w"), t("dx")


def _valid(z, name, t):
    """rows of real (not padded) positions: padded query rows are all-masked, their softmax over 's - 10000' is
    decided by fp32 rounding at that magnitude (1e-3 grid) in the reference as much as here"""
    keep = torch.from_numpy(z[f"{name}/keep"])
    return torch.as_tensor(t).detach().cpu()[keep]


def _rel(got, ref):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("name", ["sasrec", "wide"])
def test_oracle_matches_reference_modules(name):
    z, c, sd, x, mask, out, w, dx = _load(name)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xl = x.clone().requires_grad_(True)
    got = mo.encoder_forward(leaves, xl, mask, c["layers"], c["heads"], c["act"], c["eps"])
    assert _rel(_valid(z, name, got), _valid(z, name, out)) < 2e-6
    (got * w).sum().backward()
    assert _rel(xl.grad, dx) < 2e-5
    for k in leaves:
        ref = torch.from_numpy(z[f"{name}/grad/{k}"])
        g = leaves[k].grad
        if "feed_forward.LayerNorm" in k:
            assert g is None and float(ref.abs().max()) == 0.0      # created, never applied (transformer.py:116-118)
        else:
            assert _rel(g, ref) < 2e-5, k


def _build(c, sd, device):
    from gamer_amd import modules as gm
    layer = gm.TransformerEncoderLayer(c["D"], c["heads"], c["dff"], dropout=0.0, activation=c["act"],
                                       layer_norm_eps=c["eps"])
    enc = gm.TransformerEncoder(layer, c["layers"])
    assert sorted(enc.state_dict()) == sorted(sd), "state-dict keys must be the reference's"
    enc.load_state_dict(sd)
    return enc.to(device)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sasrec", "wide"])
def test_hip_encoder_matches_reference_modules(name):
    z, c, sd, x, mask, out, w, dx = _load(name)
    enc = _build(c, sd, "cuda")
    xl = x.cuda().requires_grad_(True)
    got = enc(xl, mask.cuda())
    e_out = _rel(_valid(z, name, got), _valid(z, name, out))
    (got * w.cuda()).sum().backward()
    e_dx = _rel(xl.grad, dx)
    worst, wk = 0.0, None
    for k, p in enc.named_parameters():
        ref = torch.from_numpy(z[f"{name}/grad/{k}"])
        if "feed_forward.LayerNorm" in k:
            assert p.grad is None
            continue
        if k.endswith("key.bias"):
            # softmax is invariant to a constant added to every key score: this gradient is exactly 0 in exact
            # arithmetic, the reference's value is rounding noise
            qb = dict(enc.named_parameters())[k.replace("key.bias", "query.bias")].grad
            assert float(p.grad.abs().max()) < 1e-3 * float(qb.abs().max())
            continue
        e = _rel(p.grad, ref)
        if e > worst:
            worst, wk = e, k
    assert e_out < 2e-5 and e_dx < 2e-4 and worst < 2e-4, (e_out, e_dx, worst, wk)
    # a key-padding mask of shape [B,1,1,S] broadcasts over the queries
    kp = mask[:, :, -1:, :].contiguous()
    sdc = {k: v.clone() for k, v in sd.items()}
    ref = mo.encoder_forward(sdc, x, kp, c["layers"], c["heads"], c["act"], c["eps"])
    with torch.no_grad():
        assert _rel(_valid(z, name, enc(x.cuda(), kp.cuda())), _valid(z, name, ref)) < 2e-5
        assert _rel(enc(x.cuda(), None), mo.encoder_forward(sdc, x, None, c["layers"], c["heads"], c["act"], c["eps"])) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("act", ["relu", "gelu", "swish", "tanh", "sigmoid", "elu"])
def test_bias_activation_kernels(act):
    from gamer_amd import ops
    T, N = 333, 96
    x, b, dy = torch.randn(T, N), torch.randn(N), torch.randn(T, N)
    xl, bl = x.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = mo.ACTS[act](xl + bl)
    (ref * dy.double()).sum().backward()
    xd, y = x.cuda(), torch.empty(T, N, device="cuda")
    ops.bias_act_fwd(xd, b.cuda(), ops.ACTIVATIONS[act], y)
    assert _rel(y, ref) < 2e-6 and _rel(xd, x + b) < 1e-6
    dx = torch.empty(T, N, device="cuda")
    part = torch.empty(32, N, device="cuda")
    ops.bias_act_bwd(xd, dy.cuda(), ops.ACTIVATIONS[act], dx, part)
    db = torch.empty(N, device="cuda")
    ops.colsum_reduce(part, db)
    assert _rel(dx, xl.grad) < 5e-6 and _rel(db, bl.grad) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("H", [64, 100, 1024])
def test_layernorm_kernels(H):
    from gamer_amd import ops
    T = 257
    x, r, w, b, dy = torch.randn(T, H), torch.randn(T, H), 1 + 0.1 * torch.randn(H), torch.randn(H), torch.randn(T, H)
    ls = [t.double().requires_grad_(True) for t in (x, r, w, b)]
    ref = torch.nn.functional.layer_norm(ls[0] + ls[1], (H,), ls[2], ls[3], 1e-5)
    (ref * dy.double()).sum().backward()
    f = dict(device="cuda")
    v, y, mean, rstd = torch.empty(T, H, **f), torch.empty(T, H, **f), torch.empty(T, **f), torch.empty(T, **f)
    ops.layernorm_fwd(x.cuda(), r.cuda(), w.cuda(), b.cuda(), 1e-5, v, y, mean, rstd)
    assert _rel(y, ref) < 5e-6 and _rel(v, x + r) < 1e-6
    dx, pw, pb = torch.empty(T, H, **f), torch.empty(48, H, **f), torch.empty(48, H, **f)
    ops.layernorm_bwd(v, w.cuda(), mean, rstd, dy.cuda(), dx, pw, pb)
    dw, db = torch.empty(H, **f), torch.empty(H, **f)
    ops.colsum_reduce(pw, dw)
    ops.colsum_reduce(pb, db)
    assert _rel(dx, ls[0].grad) < 2e-5 and _rel(dw, ls[2].grad) < 2e-5 and _rel(db, ls[3].grad) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,S,H,dh", [(3, 50, 2, 32), (2, 128, 3, 64), (4, 7, 1, 16)])
def test_dense_attention_with_dropout_is_consistent(B, S, H, dh):
    """Forward vs fp64 softmax attention (p = 0); with dropout: the mask takes values {0, 1/(1-p)} at the expected
    rate and the backward uses the same mask (O is linear in V: <dO, O(V+W) - O(V)> = <dV, W>)."""
    import math
    from gamer_amd import ops
    D = H * dh
    g = torch.Generator().manual_seed(S)
    q, k, v, d_o = (torch.randn(B * S, D, generator=g) for _ in range(4))
    mask = torch.where(torch.rand(B, 1, S, S, generator=g) < 0.7, 0.0, -10000.0)
    mask[:, :, torch.arange(S), torch.arange(S)] = 0.0
    ls = [t.double().requires_grad_(True) for t in (q, k, v)]
    sp = lambda t: t.view(B, S, H, dh).permute(0, 2, 1, 3)
    s = sp(ls[0]) @ sp(ls[1]).transpose(-1, -2) * math.sqrt(1.0 / dh) + mask.double()
    ref = (torch.softmax(s, -1) @ sp(ls[2])).permute(0, 2, 1, 3).reshape(B * S, D)
    (ref * d_o.double()).sum().backward()
    f = dict(device="cuda")
    qd, kd, vd, md, god = q.cuda(), k.cuda(), v.cuda(), mask.cuda(), d_o.cuda()

    def run(p, vv):
        o, lse = torch.empty(B * S, D, **f), torch.empty(B, H, S, **f)
        ops.attn_dense_fwd(qd, kd, vv, md, B, S, H, dh, math.sqrt(1.0 / dh), p, 99, o, lse)
        dq, dk, dv = (torch.empty(B * S, D, **f) for _ in range(3))
        ops.attn_dense_bwd(qd, kd, vv, md, B, S, H, dh, math.sqrt(1.0 / dh), p, 99, o, god, lse, dq, dk, dv)
        return o, dq, dk, dv
    o, dq, dk, dv = run(0.0, vd)
    assert _rel(o, ref) < 2e-5
    assert _rel(dq, ls[0].grad) < 5e-5 and _rel(dk, ls[1].grad) < 5e-5 and _rel(dv, ls[2].grad) < 5e-5
    o1, _, _, dv1 = run(0.25, vd)
    wv = torch.randn(B * S, D, generator=g).cuda()
    o2, _, _, _ = run(0.25, vd + wv)
    lhs = float(((o2 - o1).double() * god.double()).sum())
    rhs = float((dv1.double() * wv.double()).sum())
    assert abs(lhs - rhs) < 2e-4 * max(abs(rhs), 1.0)
    assert float((o1 - o).abs().max()) > 1e-3          # dropout did something
