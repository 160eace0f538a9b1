#!/usr/bin/env python3
"""Golden fixture for the evaluation path, generated from the REAL reference.

Runs ``Qwen3MultiWithTemperature.generate`` of the reference (imported from /root/reference through the shims
of ``oracle/_ref_loader.py``) exactly as ``test_SMB_decoder.py:test_single_behavior`` calls it - left-padded
history + target behaviour token, 4 new tokens, beam search constrained by the reference's ``Trie`` through
``prefix_allowed_tokens_fn_by_last_token`` - and scores the beams with the reference's ``ranking.py``.
Stores inputs, the item catalogue, the returned sequences / sequences_scores and the metrics.

``session`` argument: the same through ``Qwen3SessionMultiWithTemperature`` (test_SMB_decoder.py:181-199 passes
the collator's session_ids / extended_session_ids), histories grouped into sessions of several items.

``shipped`` argument: Qwen3MultiWithTemperature at the SHIPPED architecture (hidden 256, 8 layers, 6 / 3 heads, codebook 256,
V = 1041), max_his_len 100 (prompts of 501 tokens, one left padded), 20 beams, 2 users, target behaviours 0 (every target row
"empty") and 2 (none): BASELINE configs[4]'s shape.  The 24.5 M weights are not stored: they are ``init_state_dict(seed 17)`` with
the matrices scaled by 4 (tests/test_decode.py::_shipped_case builds the same), pinned by a checksum in the fixture.

Usage:  python oracle/make_golden_decode.py [session | shipped]   (build container only; writes tests/golden/decode_small.npz,
        decode_session_small.npz or decode_shipped.npz)
"""
import functools
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

OUT = os.path.join(ROOT, "tests", "golden", "decode_small.npz")
DIMS = dict(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1, head_dim=64,
            intermediate_size=256, moe_intermediate_size=128, behavior_embedding_dim=64,
            behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3], sparse_layers_decoder=[0, 1, 2, 3])
CB, NB, N_CAT, BEAMS, B, MAX_HIS, WSEED, WSCALE = 8, 3, 48, 6, 4, 6, 21, 4.0
METRICS = ["hit@1", "hit@5", "ndcg@5", "recall@5"]


def scaled_state_dict(ocfg):
    """init_state_dict with every matrix scaled so that the next-token distributions are peaked (beam order is
    then decided by gaps far above fp32 noise)."""
    sd = init_state_dict(ocfg, seed=WSEED)
    for k, v in sd.items():
        if v.dim() == 2:
            sd[k] = v * WSCALE
    sd["model.embed_tokens.weight"][synthetic.PAD_ID] = 0
    return sd


def main(session: bool = False):
    Model, Cfg = _ref_loader.load_reference_classes(session=session)
    out_path = OUT.replace("decode_small", "decode_session_small") if session else OUT
    from SeqRec.generation.trie import Trie, prefix_allowed_tokens_fn_by_last_token
    from SeqRec.evaluation.ranking import get_metrics_results, get_topk_results
    V = synthetic.vocab_size(CB, NB)
    bmaps = synthetic.behavior_maps(CB, NB)
    cfg = _ref_loader.reference_config(Cfg, NB, V, bmaps, n_positions=101, **DIMS)
    cfg.dropout_rate = 0.0
    cfg.attention_dropout = 0.0
    ocfg = OracleConfig.from_dict({**cfg.to_dict(), "behavior_maps": bmaps})
    sd = scaled_state_dict(ocfg)
    model = Model(cfg)
    model.set_hyper(0.7)
    model.load_state_dict({**sd, "lm_head.weight": sd["model.embed_tokens.weight"]}, strict=True)
    model.eval()
    model.generation_config.pad_token_id = synthetic.PAD_ID
    if session:
        # Shim for the installed transformers 5.x: its generate() no longer hands ``cache_position`` to forward(),
        # which Qwen3SessionMulti/model.py:969-982 relies on to advance the RoPE position of the generated tokens
        # (the pinned 4.51 passes arange(L0) for the prompt and [L0 + t] afterwards).  Supply exactly that.
        orig_forward = model.forward

        @functools.wraps(orig_forward)          # generate() validates its kwargs against forward's signature
        def forward_with_cache_position(*a, **k):
            pkv = k.get("past_key_values")
            if k.get("cache_position") is None and pkv is not None and k.get("input_ids") is not None:
                past = pkv.get_seq_length()
                k["cache_position"] = torch.arange(past, past + k["input_ids"].shape[1])
            return orig_forward(*a, **k)
        model.forward = forward_with_cache_position
    catalogue = synthetic.make_catalogue(N_CAT, CB, seed=3)
    all_item_tokens = [synthetic.item_tokens(catalogue, b, CB).tolist() for b in range(NB)]
    last_token_set = set(t[-1] for beh in all_item_tokens for t in beh)
    last_token_set.add(synthetic.PAD_ID)
    res = dict(catalogue=catalogue.numpy())
    for tb in range(NB):
        batch = synthetic.make_eval_batch(B, MAX_HIS, catalogue, tb, CB, NB, seed=40 + tb,
                                          session_mean=2.0 if session else None)
        trie = Trie(all_item_tokens[tb])
        fn = prefix_allowed_tokens_fn_by_last_token(trie, last_token_set)
        with torch.no_grad():
            out = model.generate(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                                 session_ids=batch["session_ids"], extended_session_ids=batch["extended_session_ids"],
                                 actions=batch["actions"], max_new_tokens=4, prefix_allowed_tokens_fn=fn,
                                 num_beams=BEAMS, num_return_sequences=BEAMS, output_scores=True,
                                 return_dict_in_generate=True, early_stopping=True)
        seqs, scores = out.sequences, out.sequences_scores
        # Second run with the reference's module-level cross-attention cache re-ordered together with
        # past_key_values (HF only re-orders the latter; model.py:569,785 keeps the former on the module).  Only
        # rows whose target row is "empty" can differ between the two runs.
        import transformers.cache_utils as cu
        orig_reorder = cu.DynamicCache.reorder_cache

        def reorder_both(self, beam_idx):
            orig_reorder(self, beam_idx)
            cross = model.model.cross_past_key_values
            if cross is not None and cross is not self:
                orig_reorder(cross, beam_idx)
        cu.DynamicCache.reorder_cache = reorder_both
        try:
            with torch.no_grad():
                out_fix = model.generate(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                                         session_ids=batch["session_ids"],
                                         extended_session_ids=batch["extended_session_ids"], actions=batch["actions"],
                                         max_new_tokens=4, prefix_allowed_tokens_fn=fn, num_beams=BEAMS,
                                         num_return_sequences=BEAMS, output_scores=True, return_dict_in_generate=True,
                                         early_stopping=True)
        finally:
            cu.DynamicCache.reorder_cache = orig_reorder
        res.update({f"b{tb}_sequences_crossfix": out_fix.sequences.numpy(),
                    f"b{tb}_scores_crossfix": out_fix.sequences_scores.numpy().astype(np.float64)})
        gen = seqs[:, -4:]
        pred = ["".join(f"<{int(t)}>" for t in row) for row in gen]
        tgt_tok = synthetic.item_tokens(batch["targets"], tb, CB)[:, 1:]
        # the SMB test sets carry a LIST of target items per sample (test_SMB_decoder.py:101, ranking.py:19-23)
        targets = [["".join(f"<{int(t)}>" for t in row)] for row in tgt_tok]
        topk = get_topk_results(pred, scores, targets, BEAMS)
        metrics = get_metrics_results(topk, METRICS, targets)
        gaps = (scores.view(B, BEAMS)[:, :-1] - scores.view(B, BEAMS)[:, 1:]).abs().min()
        print(f"behaviour {tb}: min score gap between ranked beams {float(gaps):.3e}; metrics {metrics}")
        res.update({f"b{tb}_input_ids": batch["input_ids"].numpy(), f"b{tb}_attention_mask": batch["attention_mask"].numpy(),
                    f"b{tb}_actions": batch["actions"].numpy(), f"b{tb}_targets": batch["targets"].numpy(),
                    f"b{tb}_session_ids": batch["session_ids"].numpy(),
                    f"b{tb}_extended_session_ids": batch["extended_session_ids"].numpy(),
                    f"b{tb}_sequences": seqs.numpy(), f"b{tb}_scores": scores.numpy().astype(np.float64),
                    f"b{tb}_topk": np.array(topk, dtype=np.int8),
                    f"b{tb}_metrics": np.array([metrics[m] for m in METRICS], dtype=np.float64)})
    meta = dict(config={**{k: getattr(ocfg, k) for k in OracleConfig.__dataclass_fields__},
                        "behavior_maps": {str(k): v for k, v in bmaps.items()}},
                model="Qwen3SessionMultiWithTemperature" if session else "Qwen3MultiWithTemperature",
                codebook=CB, num_behavior=NB, beams=BEAMS, weight_seed=WSEED, weight_scale=WSCALE, metrics=METRICS,
                generator=dict(torch=torch.__version__, transformers=__import__("transformers").__version__,
                               reference="wzf2000/GAMER @ /root/reference"))
    res["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(out_path, **res)
    print("wrote", out_path, os.path.getsize(out_path))


def main_shipped():
    """The reference's generate at the shipped dims on exactly the inputs of tests/test_decode.py::_shipped_case(B=2, tb, 11 + tb)."""
    from gamer_amd.config import synthetic_config
    Model, Cfg = _ref_loader.load_reference_classes(session=False)
    from SeqRec.generation.trie import Trie, prefix_allowed_tokens_fn_by_last_token
    cb, nb, beams, n_cat, users = 256, 3, 20, 400, 2
    scfg = synthetic_config()
    V = synthetic.vocab_size(cb, nb)
    bmaps = synthetic.behavior_maps(cb, nb)
    cfg = _ref_loader.reference_config(Cfg, nb, V, bmaps, n_positions=101)          # the reference's own config.json dims
    cfg.dropout_rate = 0.0
    cfg.attention_dropout = 0.0
    ocfg = OracleConfig.from_dict(scfg.to_dict())
    for f in ("hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim", "intermediate_size",
              "moe_intermediate_size", "behavior_embedding_dim", "vocab_size"):
        assert getattr(cfg, f) == getattr(ocfg, f), f                               # synthetic_config() IS the shipped architecture
    sd = init_state_dict(ocfg, seed=17)
    for k, v in sd.items():
        if v.dim() == 2:
            sd[k] = v * 4.0
    sd["model.embed_tokens.weight"][synthetic.PAD_ID] = 0
    model = Model(cfg)
    model.set_hyper(0.7)
    model.load_state_dict({**sd, "lm_head.weight": sd["model.embed_tokens.weight"]}, strict=True)
    model.eval()
    model.generation_config.pad_token_id = synthetic.PAD_ID
    cat = synthetic.make_catalogue(n_cat, cb, seed=3)
    all_item_tokens = [synthetic.item_tokens(cat, b, cb).tolist() for b in range(nb)]
    last_token_set = set(t[-1] for beh in all_item_tokens for t in beh)
    last_token_set.add(synthetic.PAD_ID)
    res = dict(weight_checksum=np.array([float(sum(v.double().abs().sum() for v in sd.values()))]))
    for tb in (0, 2):
        batch = synthetic.make_eval_batch(users, 100, cat, tb, codebook=cb, min_his=70, seed=11 + tb)
        fn = prefix_allowed_tokens_fn_by_last_token(Trie(all_item_tokens[tb]), last_token_set)
        with torch.no_grad():
            out = model.generate(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                                 session_ids=batch["session_ids"], extended_session_ids=batch["extended_session_ids"],
                                 actions=batch["actions"], max_new_tokens=4, prefix_allowed_tokens_fn=fn, num_beams=beams,
                                 num_return_sequences=beams, output_scores=True, return_dict_in_generate=True, early_stopping=True)
        sc = out.sequences_scores.view(users, beams)
        print(f"behaviour {tb}: prompt {tuple(batch['input_ids'].shape)}, min score gap between ranked beams "
              f"{float((sc[:, :-1] - sc[:, 1:]).abs().min()):.3e}")
        res.update({f"b{tb}_input_ids": batch["input_ids"].numpy(), f"b{tb}_attention_mask": batch["attention_mask"].numpy(),
                    f"b{tb}_actions": batch["actions"].numpy(), f"b{tb}_sequences": out.sequences.numpy(),
                    f"b{tb}_scores": out.sequences_scores.numpy().astype(np.float64)})
    meta = dict(model="Qwen3MultiWithTemperature", codebook=cb, num_behavior=nb, beams=beams, users=users, catalogue=n_cat,
                weights="oracle.qwen3multi_oracle.init_state_dict(seed=17), matrices x 4, pad row 0",
                generator=dict(torch=torch.__version__, transformers=__import__("transformers").__version__,
                               reference="wzf2000/GAMER @ /root/reference"))
    res["meta_json"] = np.array(json.dumps(meta))
    out_path = OUT.replace("decode_small", "decode_shipped")
    np.savez_compressed(out_path, **res)
    print("wrote", out_path, os.path.getsize(out_path))


if __name__ == "__main__":
    torch.set_num_threads(8)
    if "shipped" in sys.argv[1:]:
        main_shipped()
    else:
        main(session="session" in sys.argv[1:])
