#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference (build container only).

Runs the reference's ``Qwen3MultiWithTemperature`` (imported from /root/reference through the
shims in ``oracle/_ref_loader.py``) on seeded synthetic batches and stores inputs + expected
outputs.  Weights are never stored: both sides regenerate them with
``oracle.qwen3multi_oracle.init_state_dict(cfg, seed)`` and the fixture carries per-tensor fp64
checksums so a drifting generator is detected.

Gradient fixtures follow SURVEY.md section 8(a) row G1: every dropout p = 0, forward+backward
under ``sdpa_kernel(SDPBackend.MATH)`` (the reference's default CPU fused SDPA kernel produces
ill-defined gradients for fully masked rows).

The ``*_bf16`` cases run the SAME reference model under ``torch.autocast("cpu", dtype=torch.bfloat16)`` - what HF
Trainer does for the reference's ``--bf16`` flag (ref:SeqRec/tasks/train_SMB_decoder.py:114-118, 407-408) - forward
inside the autocast region, backward outside it.  They pin the AMP variant (BASELINE configs[2]).  Two facts these
fixtures establish (see oracle/qwen3multi_oracle.py::attention): the additive finfo(float32).min mask becomes -inf in
bf16 and fully masked ("empty") rows then come out as 0, not as the uniform average of the fp32 run; gradients are the
same under the default SDPA dispatch and under MATH (both recorded).

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz)
"""
import contextlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import _ref_loader  # noqa: E402
from oracle.qwen3multi_oracle import OracleConfig, init_state_dict  # noqa: E402
from gamer_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CASES = {
    # dims-generic sanity case for the oracle itself (head_dim 32)
    "tiny": dict(
        dims=dict(hidden_size=64, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1,
                  head_dim=32, intermediate_size=128, moe_intermediate_size=64, behavior_embedding_dim=32,
                  behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3],
                  sparse_layers_decoder=[0, 1, 2, 3]),
        codebook=8, B=3, n_items=7, pad_rows={1: 3}, seed=11, wseed=5, full=True),
    # smallest shape the HIP kernels support (head_dim 64, GQA group 2): full tensors
    "small": dict(
        dims=dict(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1,
                  head_dim=64, intermediate_size=256, moe_intermediate_size=128, behavior_embedding_dim=64,
                  behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3],
                  sparse_layers_decoder=[0, 1, 2, 3]),
        codebook=8, B=3, n_items=9, pad_rows={1: 4, 2: 1}, seed=12, wseed=6, full=True),
    # the shipped architecture, V=1041, S=505 (max_his_len=100), one row padded by 10 items
    "full": dict(dims=dict(), codebook=256, B=4, n_items=101, pad_rows={1: 10}, seed=1, wseed=0, full=False),
    # Qwen3SessionMulti (same weights, session-wise masks, RoPE positions = extended_session_ids): items grouped
    # into sessions of random length, raw session ids neither 0-based nor consecutive
    "session_small": dict(
        dims=dict(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1,
                  head_dim=64, intermediate_size=256, moe_intermediate_size=128, behavior_embedding_dim=64,
                  behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3],
                  sparse_layers_decoder=[0, 1, 2, 3]),
        codebook=8, B=4, n_items=15, pad_rows={1: 4, 2: 1}, seed=21, wseed=6, full=True, session_mean=2.5),
    "session_full": dict(dims=dict(), codebook=256, B=4, n_items=101, pad_rows={1: 10}, seed=2, wseed=0, full=False,
                         session_mean=4.0),
    # the reference's --bf16 run (autocast): same inputs and weights as "small" / "full"
    "small_bf16": dict(
        dims=dict(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1,
                  head_dim=64, intermediate_size=256, moe_intermediate_size=128, behavior_embedding_dim=64,
                  behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3],
                  sparse_layers_decoder=[0, 1, 2, 3]),
        codebook=8, B=3, n_items=9, pad_rows={1: 4, 2: 1}, seed=12, wseed=6, full=True, amp=True),
    "full_bf16": dict(dims=dict(), codebook=256, B=4, n_items=101, pad_rows={1: 10}, seed=1, wseed=0, full=False, amp=True),
}


def fp64_checksums(sd):
    keys = sorted(sd)
    return keys, np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())] for k in keys])


def run_case(name, spec):
    session = spec.get("session_mean") is not None
    Model, Cfg = _ref_loader.load_reference_classes(session=session)
    cb, NB = spec["codebook"], 3
    V = synthetic.vocab_size(cb, NB)
    bmaps = synthetic.behavior_maps(cb, NB)
    cfg = _ref_loader.reference_config(Cfg, NB, V, bmaps, n_positions=101, **spec["dims"])
    cfg.dropout_rate = 0.0
    cfg.attention_dropout = 0.0
    ocfg = OracleConfig.from_dict({**cfg.to_dict(), "behavior_maps": bmaps})
    sd = init_state_dict(ocfg, seed=spec["wseed"])
    model = Model(cfg)
    model.set_hyper(0.7)
    missing = model.load_state_dict({**sd, "lm_head.weight": sd["model.embed_tokens.weight"]}, strict=True)
    assert model.lm_head.weight.data_ptr() == model.model.embed_tokens.weight.data_ptr(), "head not tied"
    batch = synthetic.make_batch(spec["B"], spec["n_items"], cb, NB, seed=spec["seed"], pad_rows=spec["pad_rows"],
                                 session_mean=spec.get("session_mean"))
    fwd_in = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                  session_ids=batch["session_ids"], extended_session_ids=batch["extended_session_ids"],
                  actions=batch["actions"])
    res = {}
    amp = bool(spec.get("amp"))
    autocast = (lambda: torch.autocast("cpu", dtype=torch.bfloat16)) if amp else contextlib.nullcontext
    model.eval()
    with torch.no_grad(), autocast():
        pos, beh, act = model.model.router(batch["input_ids"].clone(),
                                           cache_position=torch.arange(batch["input_ids"].shape[1]))
        out = model(**fwd_in, output_hidden_states=True, use_cache=False)
        logits_raw = out.logits.float().clone()
        hidden = [h.float().clone() for h in out.hidden_states]
        out_l = model(**fwd_in, labels=batch["labels"], use_cache=False)
        loss_mean = float(out_l.loss)
        logits_scaled = out_l.logits.float().clone()
        n_items_tok = float((batch["labels"][:, 1:] != -100).sum()) * 2.0   # any positive number
        out_n = model(**fwd_in, labels=batch["labels"], use_cache=False, num_items_in_batch=n_items_tok)
        loss_sum = float(out_n.loss)
    # gradients: train mode, dropout 0, MATH backend
    model.train()
    from torch.nn.attention import SDPBackend, sdpa_kernel
    gnorm_default = None
    if amp:
        # what the --bf16 training run executes: default SDPA dispatch, forward under autocast, backward outside
        with autocast():
            out_d = model(**fwd_in, labels=batch["labels"], use_cache=False)
        out_d.loss.backward()
        gnorm_default = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())))
        model.zero_grad()
    with sdpa_kernel(SDPBackend.MATH):
        with autocast():
            out_g = model(**fwd_in, labels=batch["labels"], use_cache=False)
        out_g.loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    assert "lm_head.weight" not in grads or grads["lm_head.weight"] is grads.get("lm_head.weight")
    keys, sums = fp64_checksums(sd)
    gkeys = sorted(k for k in grads if k != "lm_head.weight")
    res.update(
        input_ids=batch["input_ids"].numpy(), attention_mask=batch["attention_mask"].numpy(),
        actions=batch["actions"].numpy(), labels=batch["labels"].numpy(),
        session_ids=batch["session_ids"].numpy(), extended_session_ids=batch["extended_session_ids"].numpy(),
        router_position=pos.numpy().astype(np.int16), router_behavior=beh.numpy().astype(np.int16),
        router_action=act.numpy().astype(np.int16),
        loss_mean=np.float64(loss_mean), loss_sum=np.float64(loss_sum), num_items=np.float64(n_items_tok),
        loss_train_mode=np.float64(float(out_g.loss)),
        weight_keys=np.array(keys), weight_checksums=sums,
        grad_keys=np.array(gkeys),
        grad_norms=np.array([float(grads[k].double().norm()) for k in gkeys]),
        grad_abssum=np.array([float(grads[k].double().abs().sum()) for k in gkeys]),
        global_grad_norm=np.float64(float(torch.sqrt(sum((grads[k].double() ** 2).sum() for k in gkeys)))),
        hidden_sum=np.array([float(h.double().sum()) for h in hidden]),
        hidden_abssum=np.array([float(h.double().abs().sum()) for h in hidden]),
    )
    if amp:
        res["global_grad_norm_default_sdpa"] = np.float64(gnorm_default)
        res["logits_dtype"] = np.array(str(out.logits.dtype))
    if spec["full"]:
        res["logits_raw"] = logits_raw.numpy()
        res["logits_scaled"] = logits_scaled.numpy()
        res["hidden_last"] = hidden[-1].numpy()
        res["hidden_layer1"] = hidden[1].numpy()
        for k in gkeys:
            g = grads[k]
            if g.numel() <= 20000:
                res["grad::" + k] = g.numpy()
    else:
        res["logits_raw_sample"] = logits_raw[:, ::37, ::53].numpy()
        res["logits_scaled_sample"] = logits_scaled[:, ::37, ::53].numpy()
        res["logits_raw_absmax"] = np.float64(float(logits_raw.abs().max()))
        res["hidden_last_sample"] = hidden[-1][:, ::37, ::16].numpy()
        for k in gkeys:
            g = grads[k]
            if g.dim() == 1:
                res["grad::" + k] = g.numpy()
            else:
                res["gradsample::" + k] = g[::max(1, g.shape[0] // 8), ::max(1, g.shape[1] // 8)].numpy()
    meta = dict(name=name, config={**{k: getattr(ocfg, k) for k in OracleConfig.__dataclass_fields__},
                                   "behavior_maps": {str(k): v for k, v in bmaps.items()}},
                codebook=cb, temperature=0.7, weight_seed=spec["wseed"], batch_seed=spec["seed"],
                n_items=spec["n_items"], pad_rows={str(k): v for k, v in spec["pad_rows"].items()},
                autocast="bfloat16" if amp else None,
                model="Qwen3SessionMultiWithTemperature" if session else "Qwen3MultiWithTemperature",
                generator=dict(torch=torch.__version__, transformers=__import__("transformers").__version__,
                               reference="wzf2000/GAMER @ /root/reference", sdpa_backend_for_grads="MATH"))
    res["meta_json"] = np.array(json.dumps(meta))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"{name}.npz")
    np.savez_compressed(path, **res)
    print(f"[{name}] loss_mean={loss_mean:.7f} loss_sum={loss_sum:.7f} gnorm={float(res['global_grad_norm']):.6f} "
          f"-> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or list(CASES)
    for n in which:
        run_case(n, CASES[n])
