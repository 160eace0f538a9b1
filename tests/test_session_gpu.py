"""Qwen3SessionMulti variant on the HIP path (through the C ABI): the span kernel against the reference's mask
formulas (bit-exact), the attention kernels with key spans against a dense fp64 reference, and the whole model
against the fixtures generated from the real ``Qwen3SessionMultiWithTemperature``
(``oracle/make_golden.py session_small session_full``).

Tolerances as in test_ops_gpu.py / test_model_gpu.py: attention 2e-5 (fwd) / 5e-5 (bwd) relative to the tensor's
abs-max against fp64, logits 2e-5 relative, per-tensor gradients 1e-3 relative (the north-star bar).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gamer_amd import ops, synthetic  # noqa: E402
from gamer_amd.config import Qwen3MultiConfig  # noqa: E402
from gamer_amd.engine import Engine  # noqa: E402
from oracle import qwen3multi_oracle as orc  # noqa: E402

DEV = "cuda"
INT_MAX = 0x7FFFFFFF
REPORT = {}


def _record(name, value):
    REPORT[name] = value
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "session_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


def dev(t):
    return t.to(DEV).contiguous()


def _rel(got, ref) -> float:
    got = torch.as_tensor(np.asarray(got.detach().cpu() if torch.is_tensor(got) else got)).double()
    ref = torch.as_tensor(np.asarray(ref.detach().cpu() if torch.is_tensor(ref) else ref)).double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _router_and_spans(batch, with_ext=True):
    B, S = batch["input_ids"].shape
    router = ops.alloc_router_outputs(B, S, DEV)
    lut = torch.full((64,), -1, dtype=torch.int32)
    am = dev(batch["attention_mask"])
    ops.router_fwd(dev(batch["input_ids"]), am, dev(batch["actions"]), dev(lut), 5, 4, 8, router)
    sess = ops.alloc_session_outputs(B, S, DEV)
    ops.session_spans(dev(batch["session_ids"]), dev(batch["extended_session_ids"]) if with_ext else None, am, 5, S,
                      router, sess)
    return router, sess


def _allowed_from_spans(span, kl, ql):
    """[B,S,S] bool from the kernel-side representation (what the attention kernels evaluate)."""
    span, kl = span.cpu().long(), kl.cpu().long()
    B, S = kl.shape
    j = torch.arange(S).view(1, 1, S)
    hi, lo, hh = span[..., 0:1], span[..., 1:2], span[..., 2:3]
    qlv = ql.cpu().long()[..., None] if ql is not None else torch.ones(B, S, 1, dtype=torch.long)
    return (j <= hi) & ~((j >= lo) & (j < hh)) & (kl[:, None, :] < qlv)


def _session_batch(B, n_items, seed, session_mean, pad_rows=None, left_pad=False):
    batch = synthetic.make_batch(B, n_items, 8, 3, seed=seed, pad_rows=pad_rows, session_mean=session_mean)
    if left_pad:
        # evaluation layout: the padding in front (collator.py left padding), ids/actions/sessions shifted with it
        am = batch["attention_mask"]
        for b in range(B):
            n = int(am[b].sum())
            for k in batch:
                row = batch[k][b].clone()
                batch[k][b] = torch.cat([row[n:], row[:n]])
    return batch


@pytest.mark.parametrize("session_mean,left_pad", [(1.0, False), (2.5, False), (8.0, False), (3.0, True)])
def test_session_spans_equal_reference_masks(session_mean, left_pad):
    """Integer work: bit-exact.  Spans + key/query levels must describe exactly the reference's two masks, the
    empty-row flags exactly the rows without an allowed key, pos_ids the extended session ids."""
    B, n_items = 6, 23
    batch = _session_batch(B, n_items, 100 + int(session_mean * 10), session_mean, pad_rows={0: 5, 3: 11, 4: 22},
                           left_pad=left_pad)
    S = batch["input_ids"].shape[1]
    router, sess = _router_and_spans(batch)
    self_ok, cross_ok = orc.session_mask_predicates(batch["attention_mask"], batch["actions"], batch["session_ids"], 5)
    got_self = _allowed_from_spans(sess["span_self"], router["kl_self"], None)
    got_cross = _allowed_from_spans(sess["span_cross"], router["kl_cross"], router["ql_cross"])
    assert int(sess["violations"].item()) == 0
    assert torch.equal(got_self, self_ok)
    assert torch.equal(got_cross, cross_ok)
    assert torch.equal(router["empty_self"].cpu().bool(), ~self_ok.any(-1))
    assert torch.equal(router["empty_cross"].cpu().bool(), ~cross_ok.any(-1))
    n_t = (S + 31) // 32
    pad = torch.zeros(B, n_t * 32 - S, dtype=torch.bool)
    for name, ok in (("tile_empty_self", self_ok), ("tile_empty_cross", cross_ok)):
        te = torch.cat([~ok.any(-1), pad], 1).view(B, n_t, 32).any(-1)
        assert torch.equal(router[name].cpu().bool(), te), name
    assert torch.equal(sess["pos_ids"].cpu().long(), batch["extended_session_ids"])
    assert bool((sess["span_self"][..., 0].cpu() <= torch.arange(S)).all())      # causal by construction
    assert bool((sess["span_cross"][..., 0].cpu() < torch.arange(S)).all())
    if session_mean > 1.0:
        assert bool((self_ok != orc.mask_predicates(batch["attention_mask"], batch["actions"])[0]).any())
    # without extended ids the RoPE position is the index in the sequence
    _, sess2 = _router_and_spans(batch, with_ext=False)
    assert torch.equal(sess2["pos_ids"].cpu().long(), torch.arange(S).expand(B, S))


def test_session_spans_flag_unordered_ids_and_bad_positions():
    batch = _session_batch(3, 12, 5, 2.0)
    _, sess = _router_and_spans(batch)
    assert int(sess["violations"].item()) == 0
    bad = {k: v.clone() for k, v in batch.items()}
    bad["session_ids"][1, 20:25] = 0                     # an item of a later session claims the first session
    bad["session_ids"][1, :5] = 1
    _, sess = _router_and_spans(bad)
    assert int(sess["violations"].item()) > 0
    bad = {k: v.clone() for k, v in batch.items()}
    bad["extended_session_ids"][2, 7] = 10 ** 6          # RoPE position outside the table
    _, sess = _router_and_spans(bad)
    assert int(sess["violations"].item()) == 1
    assert int(sess["pos_ids"][2, 7]) == batch["input_ids"].shape[1] - 1


def _attn_ref(q, k, v, ok, nq, nkv, scale):
    rep = nq // nkv
    kq, vq = k.repeat_interleave(rep, 2), v.repeat_interleave(rep, 2)
    s = torch.einsum("bind,bjnd->bnij", q, kq) * scale
    empty = ~ok.any(-1)
    s_eff = torch.where(empty[:, None, :, None], s - s.detach(), s.masked_fill(~ok[:, None], float("-inf")))
    p = torch.softmax(s_eff, -1)
    return torch.einsum("bnij,bjnd->bind", p, vq), torch.logsumexp(s_eff, -1), empty


@pytest.mark.parametrize("spill", [False, True, "split_h2", "split_h2_tiled"])
@pytest.mark.parametrize("use_order", [False, True])
@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("n_items,B,nq,nkv,mean", [(7, 3, 2, 1, 2.0), (14, 2, 2, 1, 3.0), (41, 3, 2, 1, 1.0),
                                                   (101, 2, 2, 1, 4.0), (101, 9, 6, 3, 6.0), (33, 40, 6, 3, 2.5),
                                                   (40, 3, 3, 3, 2.0)])
def test_session_attention_fwd_bwd(cross, n_items, B, nq, nkv, mean, use_order, spill):
    """Attention kernels with per-query key spans (SPAN instantiations) against the dense reference with the
    reference's session masks; same shapes as test_ops_gpu.test_attention_fwd_bwd plus one-item sessions.
    "split_h2": the three-product fp16 form (what the default engine runs for Qwen3SessionMulti) - since round 6 the resident kernels
    of csrc/attention_res.hip (SPAN instantiations) for the GQA group of two; "split_h2_tiled": the same call kept on the tiled
    SPAN kernels of csrc/attention_split.hip (GAMER_ATTN_RES_SPAN=0: what other group sizes and uneven CU fills run)."""
    import contextlib
    tiled = ops.env_switches(GAMER_ATTN_RES_SPAN=0) if spill == "split_h2_tiled" else contextlib.nullcontext()
    if spill == "split_h2_tiled":
        spill = "split_h2"
        if nq // nkv != 2:
            pytest.skip("only the GQA group of two has resident SPAN kernels: the plain case already ran tiled")
    with tiled:
        _session_attention_case(cross, n_items, B, nq, nkv, mean, use_order, spill, tiled=not isinstance(tiled, contextlib.nullcontext))


def _session_attention_case(cross, n_items, B, nq, nkv, mean, use_order, spill, tiled=False):
    batch = _session_batch(B, n_items, 7 + n_items, mean, pad_rows={0: max(1, n_items // 3)})
    S = batch["input_ids"].shape[1]
    T = B * S
    g = torch.Generator().manual_seed(n_items)
    q = torch.randn(B, S, nq, 64, generator=g)
    k = torch.randn(B, S, nkv, 64, generator=g)
    v = torch.randn(B, S, nkv, 64, generator=g)
    d_o = torch.randn(B, S, nq, 64, generator=g)
    self_ok, cross_ok = orc.session_mask_predicates(batch["attention_mask"], batch["actions"], batch["session_ids"], 5)
    ok = cross_ok if cross else self_ok
    leaves = [t.double().requires_grad_(True) for t in (q, k, v)]
    o_ref, lse_ref, empty = _attn_ref(*leaves, ok, nq, nkv, 0.125)
    (o_ref * d_o.double()).sum().backward()

    router, sess = _router_and_spans(batch)
    kl = router["kl_cross"] if cross else router["kl_self"]
    ql = router["ql_cross"] if cross else None
    re_ = router["empty_cross"] if cross else router["empty_self"]
    te = router["tile_empty_cross"] if cross else router["tile_empty_self"]
    span = sess["span_cross"] if cross else sess["span_self"]
    order = None
    if use_order:
        n_t = (S + 31) // 32
        order = (torch.empty(B, S, dtype=torch.int32, device=DEV), torch.empty(B, n_t, dtype=torch.int32, device=DEV),
                 torch.empty(B, n_t, dtype=torch.int32, device=DEV))
        ops.attn_row_order(re_, *order)
    ldv = (nq + 2 * nkv) * 64
    qkv = torch.zeros(T, ldv, device=DEV)
    qkv[:, (nq + nkv) * 64:] = dev(v.reshape(T, -1))
    vview = qkv[:, (nq + nkv) * 64:]
    o = torch.empty(T, nq * 64, device=DEV)
    lse = torch.empty(B, nq, S, device=DEV)
    dq_, dk_ = dev(q.reshape(T, -1)), dev(k.reshape(T, -1))
    delta = torch.empty(B, nq, S, device=DEV)
    dq = torch.empty(T, nq * 64, device=DEV)
    dk = torch.empty(T, nkv * 64, device=DEV)
    dqkv = torch.zeros(T, ldv, device=DEV)
    dvv = dqkv[:, (nq + nkv) * 64:]
    if spill == "split_h2":
        ops.attn_fwd_split(dq_, nq * 64, dk_, nkv * 64, vview, ldv, kl, ql, re_, B, S, nq, nkv, 0.125, 0.0, 1, o, lse,
                           order=order, h2=True, q_span=span)
        ops.attn_bwd_split(dq_, nq * 64, dk_, nkv * 64, vview, ldv, o, dev(d_o.reshape(T, -1)), lse, kl, ql, re_, te, B, S, nq,
                           nkv, 0.125, 0.0, 1, delta, dq, nq * 64, dk, nkv * 64, dvv, ldv, order=order, h2=True, q_span=span)
    else:
        ops.attn_fwd(dq_, nq * 64, dk_, nkv * 64, vview, ldv, kl, ql, re_, te, B, S, nq, nkv, 0.125, 0.0, 1, o, lse,
                     order=order, q_span=span)
        ds_work = torch.full((ops.attn_ds_work_numel(B, S, nq),), float("nan"), device=DEV) if spill else None
        ops.attn_bwd(dq_, nq * 64, dk_, nkv * 64, vview, ldv, o, dev(d_o.reshape(T, -1)), lse, kl, ql, re_, te, B, S, nq,
                     nkv, 0.125, 0.0, 1, delta, dq, nq * 64, dk, nkv * 64, dvv, ldv, order=order, ds_work=ds_work,
                     q_span=span)
    ne = ~empty
    e = dict(o=_rel(o, o_ref.reshape(T, -1)),
             lse=float((lse.cpu().permute(0, 2, 1)[ne].double() - lse_ref.detach().permute(0, 2, 1)[ne]).abs().max()),
             dq=_rel(dq, leaves[0].grad.reshape(T, -1)), dk=_rel(dk, leaves[1].grad.reshape(T, -1)),
             dv=_rel(dvv, leaves[2].grad.reshape(T, -1)), empty_rows=int(empty.sum()))
    _record(f"session_attn_cross{int(cross)}_S{S}_B{B}_h{nq}_ord{int(use_order)}_spill{spill}{'_tiled' if tiled else ''}", e)
    assert int((empty & batch["attention_mask"].bool()).sum()) > 0 or not cross, "fixture must contain empty rows"
    assert e["o"] < 2e-5 and e["lse"] < 2e-5
    assert e["dq"] < 5e-5 and e["dk"] < 5e-5 and e["dv"] < 5e-5


def test_session_attention_dropout_uses_the_same_mask_fwd_bwd():
    """With dropout the SPAN kernels must drop the same (row, key) pairs in the forward and in both backward
    forms: the two backward forms (recompute / dS spill) agree with each other and with a finite-difference of
    the forward along dO."""
    B, n_items, nq, nkv = 2, 14, 2, 1
    batch = _session_batch(B, n_items, 3, 2.5, pad_rows={1: 3})
    S = batch["input_ids"].shape[1]
    T = B * S
    router, sess = _router_and_spans(batch)
    g = torch.Generator().manual_seed(9)
    q, k, v = (dev(torch.randn(T, n * 64, generator=g)) for n in (nq, nkv, nkv))
    d_o = dev(torch.randn(T, nq * 64, generator=g))
    for cross in (False, True):
        kl = router["kl_cross"] if cross else router["kl_self"]
        ql = router["ql_cross"] if cross else None
        re_ = router["empty_cross"] if cross else router["empty_self"]
        te = router["tile_empty_cross"] if cross else router["tile_empty_self"]
        span = sess["span_cross"] if cross else sess["span_self"]

        def fwd(vv):
            o = torch.empty(T, nq * 64, device=DEV)
            lse = torch.empty(B, nq, S, device=DEV)
            ops.attn_fwd(q, nq * 64, k, nkv * 64, vv, nkv * 64, kl, ql, re_, te, B, S, nq, nkv, 0.125, 0.2, 77, o, lse,
                         q_span=span)
            return o, lse
        o, lse = fwd(v)
        outs = []
        for spill in (False, True):
            delta = torch.empty(B, nq, S, device=DEV)
            dq = torch.empty(T, nq * 64, device=DEV)
            dk = torch.empty(T, nkv * 64, device=DEV)
            dv = torch.empty(T, nkv * 64, device=DEV)
            ds_work = torch.empty(ops.attn_ds_work_numel(B, S, nq), device=DEV) if spill else None
            ops.attn_bwd(q, nq * 64, k, nkv * 64, v, nkv * 64, o, d_o, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, 0.2,
                         77, delta, dq, nq * 64, dk, nkv * 64, dv, nkv * 64, ds_work=ds_work, q_span=span)
            outs.append((dq, dk, dv))
        for a, b_ in zip(*outs):
            assert _rel(a, b_) < 2e-5
        # the three-product split form (csrc/attention_split.hip, SPAN instantiations) regenerates the same masks
        o_s, lse_s = torch.empty_like(o), torch.empty_like(lse)
        ops.attn_fwd_split(q, nq * 64, k, nkv * 64, v, nkv * 64, kl, ql, re_, B, S, nq, nkv, 0.125, 0.2, 77, o_s, lse_s, h2=True,
                           q_span=span)
        assert _rel(o_s, o) < 2e-5
        delta = torch.empty(B, nq, S, device=DEV)
        dq_s, dk_s, dv_s = torch.empty(T, nq * 64, device=DEV), torch.empty(T, nkv * 64, device=DEV), torch.empty(T, nkv * 64, device=DEV)
        ops.attn_bwd_split(q, nq * 64, k, nkv * 64, v, nkv * 64, o, d_o, lse, kl, ql, re_, te, B, S, nq, nkv, 0.125, 0.2, 77, delta,
                           dq_s, nq * 64, dk_s, nkv * 64, dv_s, nkv * 64, h2=True, q_span=span)
        for a, b_ in zip((dq_s, dk_s, dv_s), outs[0]):
            assert _rel(a, b_) < 5e-5
        # O is linear in V for a fixed mask: <dO, O(V + e W) - O(V)> / e = <dV, W>
        w = dev(torch.randn(T, nkv * 64, generator=g))
        o2, _ = fwd(v + 0.5 * w)
        lhs = float(((o2 - o).double() * d_o.double()).sum()) / 0.5
        rhs = float((outs[0][2].double() * w.double()).sum())
        assert abs(lhs - rhs) < 2e-4 * max(abs(rhs), 1.0), (cross, lhs, rhs)


# ---- whole model against the reference fixtures ----------------------------------------------------
def _engine_from_golden(golden, name):
    z, meta = golden(name)
    assert meta["model"] == "Qwen3SessionMultiWithTemperature"
    cfg = Qwen3MultiConfig(**meta["config"])
    cfg.dropout_rate = 0.2
    ocfg = orc.OracleConfig.from_dict(meta["config"])
    sd = orc.init_state_dict(ocfg, seed=meta["weight_seed"])
    eng = Engine(cfg, temperature=meta["temperature"], variant="session")
    eng.load_state_dict(sd)
    batch = {k: torch.from_numpy(z[k]) for k in ("input_ids", "attention_mask", "actions", "labels", "session_ids",
                                                  "extended_session_ids")}
    return z, meta, eng, batch


def _relmax(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("name", ["session_small", "session_full"])
def test_session_model_logits_and_loss_match_reference_fixture(golden, name):
    z, meta, eng, batch = _engine_from_golden(golden, name)
    skw = dict(session_ids=batch["session_ids"], extended_session_ids=batch["extended_session_ids"])
    _, logits = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], train=False, **skw)
    lg = logits.cpu().numpy()
    e_raw = _relmax(lg, z["logits_raw"]) if name == "session_small" else _relmax(lg[:, ::37, ::53], z["logits_raw_sample"])
    loss, _ = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"],
                          train=False, **skw)
    e_loss = abs(float(loss) - float(z["loss_mean"])) / float(z["loss_mean"])
    eng.check_inputs()
    _record(f"{name}_forward", dict(logits_raw=e_raw, loss=e_loss))
    assert e_raw < 2e-5
    assert e_loss < 1e-5
    with pytest.raises(ValueError):
        eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], train=False)   # ids are mandatory


@pytest.mark.parametrize("name", ["session_small", "session_full"])
def test_session_model_gradients_match_reference_fixture(golden, name):
    z, meta, eng, batch = _engine_from_golden(golden, name)
    loss, _ = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"],
                          train=True, dropout=False, session_ids=batch["session_ids"],
                          extended_session_ids=batch["extended_session_ids"])
    assert abs(float(loss) - float(z["loss_train_mode"])) < 1e-5 * float(z["loss_train_mode"])
    eng.zero_grad()
    eng.backward(1.0)
    gkeys = [str(k) for k in z["grad_keys"]]
    norms = np.array([float(eng.grads[k].double().norm()) for k in gkeys])
    rel = np.abs(norms - z["grad_norms"]) / np.maximum(z["grad_norms"], 1e-12)
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in eng.grads.values())))
    sample_err = {}
    for k in z.files:
        if k.startswith("grad::"):
            sample_err[k[6:]] = _relmax(eng.grads[k[6:]].cpu().numpy(), z[k])
        elif k.startswith("gradsample::"):
            gq = eng.grads[k[12:]]
            sample_err[k[12:]] = _relmax(gq[::max(1, gq.shape[0] // 8), ::max(1, gq.shape[1] // 8)].cpu().numpy(), z[k])
    wk = max(sample_err, key=sample_err.get)
    _record(f"{name}_gradients", dict(worst_norm_rel=float(rel.max()), worst_key=gkeys[int(rel.argmax())],
                                      worst_sample_rel=sample_err[wk], worst_sample_key=wk, global_norm=gn,
                                      global_norm_ref=float(z["global_grad_norm"])))
    assert abs(gn - float(z["global_grad_norm"])) < 1e-4 * float(z["global_grad_norm"])
    assert float(rel.max()) < 1e-3, gkeys[int(rel.argmax())]
    assert sample_err[wk] < 1e-3, wk


def test_session_train_step_matches_oracle_and_module_surface():
    """One optimizer step with the session variant (dropout off) against the oracle's step, and the nn.Module
    (Qwen3SessionMultiWithTemperature) forward/backward giving the engine's loss and gradients."""
    from gamer_amd.config import synthetic_config
    from gamer_amd.modeling import Qwen3SessionMultiWithTemperature
    cfgd = synthetic_config(256, 3).to_dict()
    cfgd.update(num_hidden_layers=4, behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3],
                sparse_layers_decoder=[0, 1, 2, 3], dropout_rate=0.0, attention_dropout=0.0)
    cfg = Qwen3MultiConfig(**cfgd)
    ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
    sd = orc.init_state_dict(ocfg, seed=4)
    batch = synthetic.make_batch(4, 19, 256, 3, ragged=True, seed=31, session_mean=3.0)
    loss_ref, grads_ref, _ = orc.loss_and_grads(sd, ocfg, batch, temperature=0.7, session=True)
    params = {k: v.clone() for k, v in sd.items() if k != "lm_head.weight"}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    vv = {k: torch.zeros_like(v) for k, v in params.items()}
    total = orc.clip_and_adamw(params, grads_ref, m, vv, 1, 1e-3)
    eng = Engine(cfg, temperature=0.7, variant="session")
    eng.load_state_dict(sd)
    loss = eng.train_step(batch, 1e-3)
    eng.check_inputs()
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * float(loss_ref)
    assert abs(float(eng.grad_norm) - float(total)) < 1e-4 * float(total)
    # Adam's first step is lr * g / (|g| + eps): compare the well-conditioned elements tightly, bound the rest by 2 lr
    # (same rule as test_model_gpu.test_gradient_accumulation_and_update_match_oracle)
    worst_p, worst_any = 0.0, 0.0
    for k in params:
        g = grads_ref[k] * float(min(1.0, 1.0 / (float(total) + 1e-6)))
        diff = (eng.params[k].cpu() - params[k]).abs()
        worst_any = max(worst_any, float(diff.max()))
        well = g.abs() > 1e-5
        if bool(well.any()):
            worst_p = max(worst_p, float(diff[well].max()))
    _record("session_train_step", dict(worst_param_well_conditioned=worst_p, worst_param_any=worst_any))
    assert worst_p < 4e-7 and worst_any <= 2 * 1e-3

    model = Qwen3SessionMultiWithTemperature(cfg)
    model.set_hyper(0.7)
    model.load_state_dict(sd)
    model.train()
    out = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"],
                session_ids=batch["session_ids"], extended_session_ids=batch["extended_session_ids"],
                actions=batch["actions"])
    assert abs(float(out.loss) - float(loss_ref)) < 1e-5 * float(loss_ref)
    out.loss.backward()
    named = dict(model.named_parameters())
    for key in ("model.layers.2.cross_attn.q_proj.weight", "model.layers.0.self_attn.k_norm.weight",
                "model.embed_tokens.weight"):
        assert _relmax(named[key].grad.cpu().numpy(), grads_ref[key].numpy()) < 1e-3, key
    bad = {k: v.clone() for k, v in batch.items()}
    bad["session_ids"] = torch.flip(bad["session_ids"], dims=[1])
    with pytest.raises(ValueError):
        model(input_ids=bad["input_ids"], attention_mask=bad["attention_mask"], session_ids=bad["session_ids"],
              extended_session_ids=bad["extended_session_ids"], actions=bad["actions"])
