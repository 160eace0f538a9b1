"""CPU oracle for GAMER's Qwen3Multi SMB-decoder train step.

TEST INFRASTRUCTURE ONLY.  This is an independent CPU (PyTorch fp32/fp64) restatement of the
reference's algorithm for the hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package ``gamer_amd`` never
does (it fails loudly when the HIP library is missing instead of falling back to this).

Pinning: ``oracle/make_golden.py`` (build container only) imports the real reference from
``/root/reference`` and writes the fixtures under ``tests/golden/``; ``tests/test_oracle.py``
checks this restatement against them (logits <= 1e-5 abs, every parameter gradient <= 1e-4
rel).  The reference ships no tests of its own for this path, so the
fixtures generated from the reference's execution are the pin.

Reference sites restated (``ref:`` = wzf2000/GAMER):
  router        ref:SeqRec/models/generative/Qwen3Multi/router.py:74-201
  masks         ref:SeqRec/models/generative/Qwen3Multi/model.py:573-630 (cross), :691-741 (self)
  attention     ref:SeqRec/models/generative/Qwen3Multi/model.py:75-150
  decoder layer ref:SeqRec/models/generative/Qwen3Multi/model.py:186-247
  sparse FFN    ref:SeqRec/models/generative/Qwen3Moe/FFN.py:25-27,53-72
  model forward ref:SeqRec/models/generative/Qwen3Multi/model.py:744-880
  head + loss   ref:SeqRec/models/generative/Qwen3Multi/model.py:904-922,928-1013
  session variant (Qwen3SessionMulti: same weights and layers, session-wise masks, RoPE positions =
                extended_session_ids)  ref:SeqRec/models/generative/Qwen3SessionMulti/model.py:545-551 (in-item mask),
                :556-613 (cross), :676-728 (self), :784-806 (both built per forward), :969-984 (positions)
  RMSNorm/RoPE/CE: third-party ``transformers`` (pinned 4.51.0 in ref:requirements.txt:9):
                models/qwen3/modeling_qwen3.py (Qwen3RMSNorm, rotate_half, apply_rotary_pos_emb,
                Qwen3RotaryEmbedding), loss/loss_utils.py (ForCausalLMLoss, fixed_cross_entropy).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


@dataclass
class OracleConfig:
    """The hyper-parameters of ref:config/s2s-models/Qwen3Multi/config.json plus the fields
    ref:SeqRec/tasks/train_SMB_decoder.py:335-360 adds at run time."""
    vocab_size: int = 1041
    hidden_size: int = 256
    num_hidden_layers: int = 8
    num_attention_heads: int = 6
    num_key_value_heads: int = 3
    head_dim: int = 64
    intermediate_size: int = 512
    moe_intermediate_size: int = 256
    behavior_embedding_dim: int = 64
    behavior_injection_decoder: List[int] = field(default_factory=lambda: [0, 1, 2, 3])
    cross_attention_decoder: List[int] = field(default_factory=lambda: [4, 5, 6, 7])
    dropout_rate: float = 0.2
    attention_dropout: float = 0.2
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    pad_token_id: int = 4
    eos_token_id: int = 8
    num_behavior: int = 3
    behavior_maps: Dict[int, int] = field(default_factory=lambda: {1038: 0, 1039: 1, 1040: 2})
    num_positions: int = 5
    num_experts: int = 6

    @staticmethod
    def from_dict(d: dict) -> "OracleConfig":
        keys = OracleConfig.__dataclass_fields__.keys()
        kw = {k: d[k] for k in keys if k in d}
        if "behavior_maps" in kw:
            kw["behavior_maps"] = {int(k): int(v) for k, v in kw["behavior_maps"].items()}
        return OracleConfig(**kw)


# --------------------------------------------------------------------------------------
# integer part
# --------------------------------------------------------------------------------------
def router(input_ids: torch.Tensor, cfg: OracleConfig):
    """router.py:74-201 for the training call (cache_position = arange(S), no user token).

    Returns (position_index, behavior_index, action_index), each [B,S] int64.
    """
    B, S = input_ids.shape
    P = cfg.num_positions
    t = torch.arange(S)
    special = (input_ids == cfg.pad_token_id) | (input_ids == cfg.eos_token_id)
    # router.py:50-58 table = (arange(P)+1).repeat(n_items) ++ [0]; :104 zero at pad/eos
    pos = ((t % P) + 1).unsqueeze(0).expand(B, S).clone()
    pos[special] = 0
    # router.py:158-195: behaviour token of each item -> map+1, repeated over the item
    first = (t // P) * P                      # index of the item's behaviour token
    beh_tok = input_ids[:, first]             # [B,S]
    act = beh_tok.clone()                     # tokens outside the map keep their raw value (:170-171)
    for tok, emb_id in cfg.behavior_maps.items():
        act[beh_tok == tok] = emb_id + 1
    act = act.clone()
    act[special] = 0
    if S % P == 1:
        # router.py:160-163 looks up (max(cache_position) + P - 1) // P = (S + 3) // 5 items: a trailing behaviour
        # token (evaluation prompt) falls on the appended "EOS" slot of router.py:176-186 -> action index 0
        act[:, -1] = 0
    beh = act.clone()
    beh[:, (t % P) == 0] = 0                  # router.py:139
    return pos, beh, act


def mask_predicates(attention_mask: torch.Tensor, actions: torch.Tensor):
    """model.py:691-741 (self) and :573-630 (cross) as boolean 'allowed' predicates [B,S,S]
    (query i, key j) plus the 'row has no allowed key' flags [B,S]."""
    B, S = attention_mask.shape
    keep = attention_mask.bool()
    i = torch.arange(S).view(1, S, 1)
    j = torch.arange(S).view(1, 1, S)
    causal = j <= i
    self_ok = causal & keep[:, None, :]
    cross_ok = causal & (actions[:, None, :] < actions[:, :, None]) & keep[:, None, :]
    return self_ok, cross_ok


def session_mask_predicates(attention_mask: torch.Tensor, actions: torch.Tensor, session_ids: torch.Tensor, P: int):
    """Qwen3SessionMulti masks as 'allowed' predicates [B,S,S] (query i, key j).

    self  (model.py:676-728 ``_update_session_wise_causal_mask``): finfo.min * in_item_mask * (sess_j >= sess_i),
          in_item_mask = 1 - (I + strictly-lower triangle inside every block of P positions) (model.py:545-551):
          a key is visible when it is the query itself or an earlier token of the query's item, or when it
          belongs to a strictly earlier session; other items of the query's own session are hidden.
    cross (model.py:556-613): masked when sess_j >= sess_i or act_j >= act_i.
    Both then mask padded keys (allowed & attention_mask == 0 -> finfo.min).  Nothing else enforces causality:
    it follows from session ids that do not decrease along the sequence."""
    B, S = attention_mask.shape
    keep = attention_mask.bool()
    i = torch.arange(S).view(1, S, 1)
    j = torch.arange(S).view(1, 1, S)
    in_item = (torch.div(j, P, rounding_mode="floor") == torch.div(i, P, rounding_mode="floor")) & (j <= i)
    earlier = session_ids[:, None, :] < session_ids[:, :, None]
    self_ok = (in_item | earlier) & keep[:, None, :]
    cross_ok = earlier & (actions[:, None, :] < actions[:, :, None]) & keep[:, None, :]
    return self_ok, cross_ok


# --------------------------------------------------------------------------------------
# float part
# --------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen3RMSNorm.forward: fp32 variance, cast back, then multiply by the weight."""
    dt = x.dtype
    xf = x.to(torch.float32) if dt != torch.float64 else x
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def rope_tables(S: int, dh: int, theta: float, dtype=torch.float32):
    """Qwen3RotaryEmbedding.forward with position_ids = arange(S) (model.py:787-794,819)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.int64).to(torch.float32) / dh))
    pos = torch.arange(S, dtype=torch.float32)
    freqs = pos[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x, cos, sin):
    """x [B,S,heads,dh]; cos/sin [S,dh] (positions = arange) or [B,S,dh] (per-token positions)."""
    if cos.dim() == 3:
        return x * cos[:, :, None, :] + rotate_half(x) * sin[:, :, None, :]
    return x * cos[None, :, None, :] + rotate_half(x) * sin[None, :, None, :]


def _dropout(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0) else x


BF16 = torch.bfloat16


def _linear(x, w, amp: bool):
    """nn.Linear; under autocast (``amp``) both operands are cast to bf16 and the result is bf16 (fp32 accumulation
    inside the matmul, one rounding on the way out)."""
    if amp:
        return F.linear(x.to(BF16), w.to(BF16))
    return F.linear(x, w)


def attention(h, sd, prefix, ok, cfg: OracleConfig, cos, sin, act_idx=None, training=False, uniform_len=None,
              amp: bool = False, o_hook=None):
    """Qwen3MultiAttention.forward (model.py:75-150) with the additive finfo.min mask folded in.

    ``amp`` = the reference under ``torch.autocast(bfloat16)`` (its ``--bf16`` run, train_SMB_decoder.py:114-118,
    407-408), restated with explicit casts: projections in bf16; q/k-norm in fp32 on what it is handed (a bf16
    tensor for the self attention: the normalised value is rounded to bf16 before the fp32 weight multiply; an fp32
    tensor for the cross attention, where the fp32 behaviour embedding was added first); RoPE in fp32; SDPA's q, k, v
    AND its additive mask are cast to bf16 - finfo(float32).min is not representable and becomes -inf, and SDPA's
    softmax returns 0 for a row whose scores are all -inf.  So under bf16 an "empty" query row contributes NOTHING
    (output 0, no gradient) instead of the uniform average over all keys of the fp32 path.  Verified against the
    reference executed under autocast on CPU (oracle/make_golden.py, fixtures *_bf16).

    A query row with no allowed key ends up with every masked score equal to finfo.min, so the
    softmax is uniform over all S keys (future and padded ones included) while autograd still
    passes d(score) through the addition; ``s - s.detach()`` reproduces exactly that.

    ``uniform_len`` (evaluation by re-running the whole sequence): in the reference's cached generation an
    empty row is uniform over the keys that existed WHEN THE ROW WAS COMPUTED - the prompt length L0 for a
    prompt row, i+1 for a generated row i (model.py:603-617: the cached last mask row grows by one masked
    key per step) - so row i spans max(uniform_len, i+1) keys.  None = all S keys (training).
    """
    B, S, _ = h.shape
    nq, nkv, dh = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    cross = act_idx is not None
    q = _linear(h, sd[prefix + "q_proj.weight"], amp).view(B, S, nq, dh)
    k = _linear(h, sd[prefix + "k_proj.weight"], amp).view(B, S, nkv, dh)
    v = _linear(h, sd[prefix + "v_proj.weight"], amp).view(B, S, nkv, dh)
    if cross:
        q = q + sd[prefix + "q_behavior_embedding.weight"][act_idx].view(B, S, nq, dh)
        k = k + sd[prefix + "k_behavior_embedding.weight"][act_idx].view(B, S, nkv, dh)
        v = v + sd[prefix + "v_behavior_embedding.weight"][act_idx].view(B, S, nkv, dh)
    q = apply_rope(rmsnorm(q, sd[prefix + "q_norm.weight"], cfg.rms_norm_eps), cos, sin)
    k = apply_rope(rmsnorm(k, sd[prefix + "k_norm.weight"], cfg.rms_norm_eps), cos, sin)
    rep = nq // nkv
    kq = k.repeat_interleave(rep, dim=2)          # query head n uses kv head n // rep
    vq = v.repeat_interleave(rep, dim=2)
    empty = ~ok.any(-1)                           # [B,S]
    okb = ok[:, None, :, :]
    if amp:
        # SDPA under autocast: bf16 operands, fp32 scores / softmax statistics, un-normalised probabilities rounded to
        # bf16 for the second product (what the fused kernels do), bf16 output; all -inf rows -> 0
        qb, kb, vb = q.to(BF16).float(), kq.to(BF16).float(), vq.to(BF16).float()
        s = torch.einsum("bind,bjnd->bnij", qb, kb) * (dh ** -0.5)
        s = s.masked_fill(~okb, float("-inf"))
        m = s.amax(-1, keepdim=True)
        m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
        e = torch.exp(s - m)                                     # masked -> 0
        l = e.sum(-1, keepdim=True)
        e = _dropout(e, cfg.attention_dropout, training)
        o = torch.einsum("bnij,bjnd->bind", e.to(BF16).float(), vb) / l.clamp_min(1e-30).transpose(1, 2)
        o = torch.where(empty[:, :, None, None], torch.zeros_like(o), o).to(BF16).reshape(B, S, nq * dh)
        out = _linear(o, sd[prefix + "o_proj.weight"], amp)
        if cross:
            out = out * F.silu(_linear(h, sd[prefix + "gating.weight"], amp))
        return out
    s = torch.einsum("bind,bjnd->bnij", q, kq) * (dh ** -0.5)
    s_norm = s.masked_fill(~okb, float("-inf"))
    s_empty = s - s.detach()
    if uniform_len is not None:
        span = torch.clamp(torch.arange(S) + 1, min=int(uniform_len))                  # [S] keys seen by row i
        s_empty = s_empty.masked_fill(torch.arange(S)[None, :] >= span[:, None], float("-inf"))
    s_eff = torch.where(empty[:, None, :, None], s_empty, s_norm)
    p = torch.softmax(s_eff, dim=-1)
    p = _dropout(p, cfg.attention_dropout, training)
    o = torch.einsum("bnij,bjnd->bind", p, vq)
    if o_hook is not None:
        # evaluation only (oracle/decode_oracle.py): lets the decode restatement replace attention outputs whose
        # VALUE rows come from the reference's un-reordered cross-attention cache
        o = o_hook(prefix, o, v, empty)
    o = o.reshape(B, S, nq * dh)
    out = F.linear(o, sd[prefix + "o_proj.weight"])
    if cross:
        out = out * F.silu(F.linear(h, sd[prefix + "gating.weight"]))
    return out


def sparse_mlp(h, sd, prefix, pos_idx, beh_idx, cfg: OracleConfig, inject: bool, training=False, amp: bool = False):
    """MyQwen3SparseMLP.forward (FFN.py:53-72): every token goes through exactly one expert,
    chosen by its position index; layers in behavior_injection_decoder concatenate a
    behaviour embedding first."""
    if inject:
        h = torch.cat((h, sd[prefix + "behavior_embedding.weight"][beh_idx]), dim=-1)
    out = torch.zeros(h.shape[:-1] + (cfg.moe_intermediate_size,), dtype=h.dtype)
    for e in range(cfg.num_experts):
        sel = pos_idx == e
        if not bool(sel.any()):
            continue
        x = h[sel]
        ep = f"{prefix}experts.expert_{e}."
        g = _linear(x, sd[ep + "gate_proj.weight"], amp)
        u = _linear(x, sd[ep + "up_proj.weight"], amp)
        m = _dropout(F.silu(g) * u, cfg.dropout_rate, training)
        out[sel] = _linear(m, sd[ep + "down_proj.weight"], amp).to(out.dtype)      # FFN.py:66-68
    return out


def forward(sd: Dict[str, torch.Tensor], cfg: OracleConfig, input_ids, attention_mask, actions,
            labels=None, temperature: float = 1.0, num_items_in_batch: Optional[float] = None,
            training: bool = False, return_hidden: bool = False, act_zero_col: Optional[int] = None,
            uniform_len: Optional[int] = None, session_ids=None, extended_session_ids=None, amp: bool = False,
            cross_o_hook=None):
    """Qwen3MultiWithTemperature.forward (model.py:928-1013).

    ``session_ids`` given: the Qwen3SessionMulti variant (``session_mask_predicates``; with
    ``extended_session_ids`` as the RoPE positions, Qwen3SessionMulti/model.py:983-984).

    ``act_zero_col``: evaluation prompts end with the target item's behaviour token (S = 5n+1); the reference's
    router then looks up only n items (router.py:160-163 with cache_position) and that token gets action index 0,
    which stays in the cross-attention K/V cache for the whole generation.  Passing the column reproduces it.
    ``uniform_len``: see ``attention``.

    ``amp``: the reference under ``torch.autocast(bfloat16)`` (see ``attention``); parameters, the residual stream,
    the norms and the loss stay fp32, the logits come out in bf16 and ``logits /= T`` rounds them once more.

    ``sd`` uses the reference's state-dict key names.  Returns a dict with ``logits`` (divided
    by the temperature when labels are given, as the reference's in-place ``logits /= T`` does),
    ``loss`` (or None), and optionally the per-layer hidden states and the router maps.
    """
    B, S = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    dtype = sd["model.embed_tokens.weight"].dtype
    pos_idx, beh_idx, act_idx = router(input_ids, cfg)
    if act_zero_col is not None:
        act_idx = act_idx.clone()
        act_idx[:, act_zero_col] = 0
    if session_ids is not None:
        self_ok, cross_ok = session_mask_predicates(attention_mask, actions, session_ids, cfg.num_positions)
    else:
        self_ok, cross_ok = mask_predicates(attention_mask, actions)
    if extended_session_ids is not None:
        n_pos = max(S, int(extended_session_ids.max()) + 1)
        cos, sin = rope_tables(n_pos, cfg.head_dim, cfg.rope_theta, dtype)
        cos, sin = cos[extended_session_ids], sin[extended_session_ids]       # [B,S,dh]
    else:
        cos, sin = rope_tables(S, cfg.head_dim, cfg.rope_theta, dtype)
    # nn.Embedding(vocab, H, padding_idx=pad) (model.py:263): the gather-side gradient of the pad row is dropped
    x = F.embedding(input_ids, sd["model.embed_tokens.weight"], padding_idx=cfg.pad_token_id)
    hidden = [] if return_hidden else None      # model.py:822-873: input of every layer + final norm
    eps = cfg.rms_norm_eps
    for l in range(cfg.num_hidden_layers):
        lp = f"model.layers.{l}."
        if return_hidden:
            hidden.append(x)
        h = rmsnorm(x, sd[lp + "input_layernorm.weight"], eps)
        a = attention(h, sd, lp + "self_attn.", self_ok, cfg, cos, sin, None, training, uniform_len, amp)
        x = x + _dropout(a, cfg.dropout_rate, training)
        if l in cfg.cross_attention_decoder:
            h = rmsnorm(x, sd[lp + "post_self_attention_layernorm.weight"], eps)
            a = attention(h, sd, lp + "cross_attn.", cross_ok, cfg, cos, sin, act_idx, training, uniform_len, amp,
                          o_hook=cross_o_hook)
            x = x + _dropout(a, cfg.dropout_rate, training)
        h = rmsnorm(x, sd[lp + "post_cross_attention_layernorm.weight"], eps)
        m = sparse_mlp(h, sd, lp + "mlp.", pos_idx, beh_idx, cfg,
                       l in cfg.behavior_injection_decoder, training, amp)
        x = x + _dropout(m, cfg.dropout_rate, training)
    xn = rmsnorm(x, sd["model.norm.weight"], eps)
    if return_hidden:
        hidden.append(xn)
    head = sd.get("lm_head.weight", sd["model.embed_tokens.weight"])
    logits = _linear(xn, head, amp)
    loss = None
    if labels is not None:
        logits = logits / temperature                      # model.py:913 (in place upstream)
        shift = F.pad(labels, (0, 1), value=-100)[:, 1:]    # loss_utils.py: shift left, pad -100
        flat = logits.reshape(-1, logits.shape[-1]).float() if dtype != torch.float64 \
            else logits.reshape(-1, logits.shape[-1])
        tgt = shift.reshape(-1)
        if num_items_in_batch is not None:
            loss = F.cross_entropy(flat, tgt, ignore_index=-100, reduction="sum") / num_items_in_batch
        else:
            loss = F.cross_entropy(flat, tgt, ignore_index=-100, reduction="mean")
    out = {"logits": logits, "loss": loss, "router": (pos_idx, beh_idx, act_idx)}
    if return_hidden:
        out["hidden_states"] = hidden
    return out


# --------------------------------------------------------------------------------------
# parameters / update (for gradient parity and the CPU baseline)
# --------------------------------------------------------------------------------------
def param_shapes(cfg: OracleConfig) -> Dict[str, tuple]:
    """Key -> shape of every distinct parameter (tied lm_head omitted), in the reference's
    state-dict order (model.py:38-66,161-176,263-285; FFN.py:19-21,42-51)."""
    H, dh = cfg.hidden_size, cfg.head_dim
    nq, nkv, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    Eb, NB = cfg.behavior_embedding_dim, cfg.num_behavior
    shapes = {"model.embed_tokens.weight": (cfg.vocab_size, H)}
    for l in range(cfg.num_hidden_layers):
        lp = f"model.layers.{l}."
        cross = l in cfg.cross_attention_decoder
        inject = l in cfg.behavior_injection_decoder
        for a in (["self_attn", "cross_attn"] if cross else ["self_attn"]):
            ap = lp + a + "."
            shapes[ap + "q_proj.weight"] = (nq * dh, H)
            shapes[ap + "k_proj.weight"] = (nkv * dh, H)
            shapes[ap + "v_proj.weight"] = (nkv * dh, H)
            shapes[ap + "o_proj.weight"] = (H, nq * dh)
            shapes[ap + "q_norm.weight"] = (dh,)
            shapes[ap + "k_norm.weight"] = (dh,)
            if a == "cross_attn":
                shapes[ap + "q_behavior_embedding.weight"] = (NB + 1, nq * Eb)
                shapes[ap + "k_behavior_embedding.weight"] = (NB + 1, nkv * Eb)
                shapes[ap + "v_behavior_embedding.weight"] = (NB + 1, nkv * Eb)
                shapes[ap + "gating.weight"] = (H, H)
        if cross:
            shapes[lp + "post_self_attention_layernorm.weight"] = (H,)
        din = cfg.moe_intermediate_size + (Eb if inject else 0)
        for e in range(cfg.num_experts):
            ep = f"{lp}mlp.experts.expert_{e}."
            shapes[ep + "gate_proj.weight"] = (I, din)
            shapes[ep + "up_proj.weight"] = (I, din)
            shapes[ep + "down_proj.weight"] = (cfg.moe_intermediate_size, I)
        if inject:
            shapes[lp + "mlp.behavior_embedding.weight"] = (NB + 1, Eb)
        shapes[lp + "input_layernorm.weight"] = (H,)
        shapes[lp + "post_cross_attention_layernorm.weight"] = (H,)
    shapes["model.norm.weight"] = (H,)
    return shapes


def init_state_dict(cfg: OracleConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic fill shared by the fixtures and the tests: parameters visited in sorted key
    order, normal(0, 0.02) from one seeded generator; RMSNorm weights are 1 + normal(0, 0.1)
    (non-trivial so that norm-weight handling is actually exercised); padding row of the
    embedding zeroed.  This is NOT the HF initialiser (SURVEY.md section 8(a) note) — parity
    tests always load explicit weights."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    shapes = param_shapes(cfg)
    for k in sorted(shapes):
        shp = shapes[k]
        if len(shp) == 1:
            sd[k] = (1.0 + 0.1 * torch.randn(shp, generator=g, dtype=torch.float64)).to(dtype)
        else:
            sd[k] = (0.02 * torch.randn(shp, generator=g, dtype=torch.float64)).to(dtype)
    sd["model.embed_tokens.weight"][cfg.pad_token_id].zero_()
    return sd


NO_DECAY_SUFFIXES = ("layernorm.weight", "_norm.weight", "model.norm.weight")


def is_no_decay(key: str) -> bool:
    """HF Trainer.get_decay_parameter_names: RMSNorm/LayerNorm weights and biases get no weight
    decay (transformers/trainer.py get_decay_parameter_names; trainer_pt_utils.get_parameter_names
    with forbidden 'norm' name patterns).  Embeddings and linear weights are decayed."""
    return key.endswith(NO_DECAY_SUFFIXES)


def loss_and_grads(sd, cfg, batch, temperature=1.0, num_items_in_batch=None, training=False, session=False,
                   amp: bool = False):
    """Forward + autograd backward; returns (loss, {key: grad}, forward-output) with the tied
    table's gradient under 'model.embed_tokens.weight' (head wgrad over all rows + gather
    scatter-add with the padding row's contribution dropped, as nn.Embedding(padding_idx) does)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if k != "lm_head.weight"}
    view = dict(leaves)
    view["lm_head.weight"] = leaves["model.embed_tokens.weight"]
    out = forward(view, cfg, batch["input_ids"], batch.get("attention_mask"), batch["actions"],
                  labels=batch.get("labels"), temperature=temperature,
                  num_items_in_batch=num_items_in_batch, training=training, amp=amp,
                  session_ids=batch["session_ids"] if session else None,
                  extended_session_ids=batch["extended_session_ids"] if session else None)
    out["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return out["loss"].detach(), grads, out


def clip_and_adamw(params: Dict[str, torch.Tensor], grads, m, v, step: int, lr: float,
                   beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, max_norm=1.0):
    """HF Trainer update (transformers/trainer.py: clip_grad_norm_(max_norm) then
    torch.optim.AdamW.step; ref:SeqRec/tasks/train_SMB_decoder.py:396-428 picks adamw_torch).
    ``step`` is 1-based.  Updates params/m/v in place, returns the pre-clip global grad norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k, p in params.items():
        g = grads[k] * coef
        wd = 0.0 if is_no_decay(k) else weight_decay
        p.mul_(1.0 - lr * wd)
        m[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
        v[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m[k], denom, value=-lr / bc1)
    return total


def cosine_lr(step: int, base_lr: float, warmup_steps: int, total_steps: int) -> float:
    """transformers.get_cosine_schedule_with_warmup (HF Trainer default for lr_scheduler_type
    'cosine', ref:SeqRec/tasks/train_SMB_decoder.py:396-428); ``step`` = number of optimizer
    steps already taken."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    prog = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
