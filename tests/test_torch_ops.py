"""torch.ops.gamer.* (TORCH_LIBRARY registration over the C ABI, csrc/torch_ops.cpp).
CPU: the library loads, every op is registered with its schema, CPU tensors are refused by the dispatcher.
GPU: each op against the same entry point driven through ctypes (gamer_amd.ops) - identical bits - and the module
path's FusedClipAdamW (which runs through gamer::fused_adamw_clip) against Engine.optimizer_step."""
import pytest
import torch

from gamer_amd import torch_ops


def test_ops_are_registered_with_schemas():
    ns = torch_ops.load()
    for name in torch_ops.OPS:
        schema = str(getattr(ns, name).default._schema)
        assert schema.startswith(f"gamer::{name}("), schema
    assert "Tensor(a!) logits" in str(ns.lmhead_ce_fwd.default._schema)       # in-place /temperature, as upstream


def test_cpu_tensors_are_refused():
    ns = torch_ops.load()
    with pytest.raises((NotImplementedError, RuntimeError)):
        ns.rmsnorm_fwd(torch.randn(4, 64), torch.ones(64), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_ops_match_the_ctypes_path(dt):
    from gamer_amd import ops, synthetic
    ns = torch_ops.load()
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    B, n_items, nq, nkv, H = 3, 9, 2, 1, 128
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=4, pad_rows={1: 2})
    S = batch["input_ids"].shape[1]
    T = B * S
    bf = dt == torch.bfloat16
    x, w = torch.randn(T, H, generator=g).to(dev), (torch.rand(H, generator=g) + 0.5).to(dev)
    # rmsnorm fwd / bwd
    y = ns.rmsnorm_fwd(x, w, 1e-6, bf)
    y_ref = torch.empty(T, H, dtype=dt, device=dev)
    ops.rmsnorm_fwd(x, w, 1e-6, y_ref)
    assert torch.equal(y, y_ref)
    dy = torch.randn(T, H, generator=g).to(dev).to(dt)
    dx, dw = ns.rmsnorm_bwd(x, w, dy, 1e-6)
    dx_ref, part, dw_ref = torch.zeros(T, H, device=dev), torch.empty(512, H, device=dev), torch.zeros(H, device=dev)
    ops.rmsnorm_bwd(x, w, dy, H, 1e-6, dx_ref, part, False)
    ops.colsum_reduce(part, dw_ref)
    assert torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref)
    # linear
    K, N = (64, 96) if bf else (40, 52)
    a, wt = torch.randn(T, K, generator=g).to(dev).to(dt), torch.randn(N, K, generator=g).to(dev).to(dt)
    out = ns.linear(a, wt)
    out_ref = torch.empty(T, N, dtype=dt, device=dev)
    ops.linear_fwd(a, K, wt, K, out_ref, N, T, N, K)
    assert torch.equal(out, out_ref)
    ref_lin = a.float() @ wt.float().T
    assert float((out.float() - ref_lin).abs().max()) < (1e-2 if bf else 1e-5) * float(ref_lin.abs().max())
    # qkv norm + rope, attention fwd / bwd (self: ql undefined)
    from oracle import qwen3multi_oracle as orc
    cos, sin = (t.to(dev) for t in orc.rope_tables(S, 64, 1e6))
    qkv = torch.randn(T, (nq + 2 * nkv) * 64, generator=g).to(dev).to(dt)
    wq, wk = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.rand(64, generator=g) + 0.5).to(dev)
    qkv2 = qkv.clone()
    q, k = ns.qkv_rope_fwd(qkv, S, nq, nkv, wq, wk, 1e-6, cos, sin)
    q_ref, k_ref = torch.empty_like(q), torch.empty_like(k)
    ops.qknorm_rope_fwd(qkv2, S, nq, nkv, wq, wk, 1e-6, cos, sin, q_ref, k_ref)
    assert torch.equal(q, q_ref) and torch.equal(k, k_ref)
    router = ops.alloc_router_outputs(B, S, dev)
    lut = torch.full((64,), -1, dtype=torch.int32, device=dev)
    ops.router_fwd(batch["input_ids"].to(dev), batch["attention_mask"].to(dev), batch["actions"].to(dev), lut, 5, 4, 8, router)
    v = qkv[:, (nq + nkv) * 64:]
    o, lse = ns.mb_attention_fwd(q, k, v, router["kl_self"], None, router["empty_self"], router["tile_empty_self"], B, S, nq, nkv,
                                 0.125, 0.2, 77)
    o_ref, lse_ref = torch.empty_like(o), torch.empty_like(lse)
    if bf:
        ops.attn_fwd_bf16(q, nq * 64, k, nkv * 64, v, v.stride(0), router["kl_self"], None, B, S, nq, nkv, 0.125, 0.2, 77, o_ref, lse_ref)
    else:
        ops.attn_fwd(q, nq * 64, k, nkv * 64, v, v.stride(0), router["kl_self"], None, router["empty_self"],
                     router["tile_empty_self"], B, S, nq, nkv, 0.125, 0.2, 77, o_ref, lse_ref)
    assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref)
    d_o = torch.randn(T, nq * 64, generator=g).to(dev).to(dt)
    dq, dk, dv = ns.mb_attention_bwd(q, k, v, o, d_o, lse, router["kl_self"], None, router["empty_self"], router["tile_empty_self"],
                                     B, S, nq, nkv, 0.125, 0.2, 77)
    assert all(bool(torch.isfinite(t.float()).all()) for t in (dq, dk, dv)) and float(dq.float().abs().max()) > 0
    # swiglu
    gg, uu, dh = (torch.randn(T, 256, generator=g).to(dev).to(dt) for _ in range(3))
    hm = ns.swiglu_fwd(gg, uu, 0.2, 5)
    hm_ref = torch.empty_like(hm)
    ops.swiglu_fwd(gg, uu, T * 256, 0.2, 5, hm_ref)
    assert torch.equal(hm, hm_ref)
    dg, du = ns.swiglu_bwd(gg, uu, dh, 0.2, 5)
    g2, u2 = gg.clone(), uu.clone()
    ops.swiglu_bwd(g2, u2, dh, T * 256, 0.2, 5)
    assert torch.equal(dg, g2) and torch.equal(du, u2)
    # head loss
    V, ld = 50, 64
    lg = (torch.randn(T, ld, generator=g) * 2).to(dev).to(dt)
    lg[:, V:] = 0
    labels = torch.randint(0, V, (B, S), generator=g).to(dev)
    lg2 = lg.clone()
    tot, cnt, lse_ce = ns.lmhead_ce_fwd(lg, labels, V, 0.7)
    shift = torch.nn.functional.pad(labels, (0, 1), value=-100)[:, 1:].reshape(-1)
    zs = (lg2[:, :V].float() / 0.7).to(dt).float()
    ref = torch.nn.functional.cross_entropy(zs, shift, ignore_index=-100, reduction="sum")
    assert abs(float(tot) - float(ref)) < 1e-4 * float(ref) and float(cnt) == float((shift != -100).sum())
    ns.lmhead_ce_bwd(lg, labels, lse_ce, V, 0.7, cnt, torch.ones(1, device=dev))
    zz = zs.clone().requires_grad_(True)
    torch.nn.functional.cross_entropy(zz, shift, ignore_index=-100, reduction="mean").backward()
    assert float((lg[:, :V].float() - zz.grad / 0.7).abs().max()) <= (2 ** -8 if bf else 1e-6) * float((zz.grad / 0.7).abs().max()) + 1e-9


@pytest.mark.gpu
def test_fused_optimizer_runs_through_the_registered_op():
    """module.fused_optimizer() (gamer::fused_adamw_clip) takes the same step as Engine.optimizer_step."""
    from gamer_amd import synthetic
    from gamer_amd.config import synthetic_config
    from gamer_amd.engine import Engine
    from gamer_amd.modeling import Qwen3MultiWithTemperature
    cfg = synthetic_config(codebook=8, num_hidden_layers=2, behavior_injection_decoder=[0], cross_attention_decoder=[1])
    cfg.dropout_rate = cfg.attention_dropout = 0.0
    model = Qwen3MultiWithTemperature(cfg)
    model.set_hyper(0.7)
    model.train()
    eng = Engine(cfg, temperature=0.7)
    eng.load_state_dict({k: v.clone() for k, v in model.engine.params.items()})
    opt = model.fused_optimizer(lr=1e-3)
    p0 = eng.flat_p.clone()
    for step in range(3):
        b = synthetic.make_batch(4, 6, 8, 3, seed=step)
        bd = {k: v.cuda() for k, v in b.items()}
        out = model(input_ids=bd["input_ids"], attention_mask=bd["attention_mask"], actions=bd["actions"], labels=bd["labels"])
        out.loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        eng.train_step(b, 1e-3)
    # two runs of the same step differ by the summation order of the split-K fp32 atomics (and Adam turns a sign flip
    # of a ~0 gradient into a full lr-sized difference on that element): compare the UPDATES in norm
    upd = (eng.flat_p - p0).double()
    assert float((model.engine.flat_p - eng.flat_p).double().norm() / upd.norm()) < 2e-2
    assert abs(float(model.engine.grad_norm) - float(eng.grad_norm)) < 1e-4 * float(eng.grad_norm)
    sd = opt.state_dict()
    assert sd["state"]["step"] == 3 and sd["state"]["m"] is not None


@pytest.mark.gpu
def test_routed_swiglu_ops_against_dense_reference():
    """gamer::routed_swiglu_fwd / _bwd (FFN.py:53-72 in expert-sorted order) against a per-expert fp64 loop."""
    ns = torch_ops.load()
    dev = "cuda"
    g = torch.Generator().manual_seed(3)
    E, I, din = 6, 128, 96
    sizes = [0, 130, 257, 1, 64, 200]                       # expert 0 empty (the pad / eos expert), ragged segments
    offs = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32)
    T = int(offs[-1])
    hin = torch.randn(T, din, generator=g)
    wg, wu = torch.randn(E * I, din, generator=g) * 0.1, torch.randn(E * I, din, generator=g) * 0.1
    dhm = torch.randn(T, I, generator=g)
    gt, ut, hm = ns.routed_swiglu_fwd(hin.to(dev), wg.to(dev), wu.to(dev), offs.to(dev))
    leaves = [t.double().requires_grad_(True) for t in (hin, wg, wu)]
    outs = []
    for e in range(E):
        a, b = int(offs[e]), int(offs[e + 1])
        x = leaves[0][a:b]
        ge, ue = x @ leaves[1][e * I:(e + 1) * I].T, x @ leaves[2][e * I:(e + 1) * I].T
        outs.append(torch.nn.functional.silu(ge) * ue)
    ref = torch.cat(outs)
    (ref * dhm.double()).sum().backward()
    rel = lambda got, want: float((got.cpu().double() - want).abs().max() / want.abs().max())
    assert rel(hm, ref.detach()) < 2e-6
    dhin, dwg, dwu = ns.routed_swiglu_bwd(gt, ut, dhm.to(dev), hin.to(dev), wg.to(dev), wu.to(dev), offs.to(dev))
    assert rel(dhin, leaves[0].grad) < 5e-6 and rel(dwg, leaves[1].grad) < 5e-6 and rel(dwu, leaves[2].grad) < 5e-6
    assert float(dwg[:I].abs().max()) == 0.0                 # the empty expert gets a zero gradient


def test_allreduce_bucket_is_the_identity_without_a_group():
    ns = torch_ops.load()
    f = torch.arange(8.0)
    ns.allreduce_bucket(f, 2, 5)
    assert torch.equal(f, torch.arange(8.0))
    with pytest.raises(RuntimeError):
        ns.allreduce_bucket(f, 5, 2)
