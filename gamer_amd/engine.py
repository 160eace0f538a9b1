"""Train-step engine for the Qwen3Multi SMB decoder on MI355X.

Host-side orchestration of the HIP kernels behind the C ABI (``gamer_amd.ops``): explicit
forward and backward chains over pre-allocated HBM buffers, flat fp32 parameter / gradient /
Adam-moment buffers, fused clip+AdamW.  No autograd, no host synchronisation in the steady state
(the reference has >= 48 host syncs per forward from its boolean-mask expert loop,
ref:SeqRec/models/generative/Qwen3Moe/FFN.py:63-68).

What it computes is Appendix A of SURVEY.md, i.e.
  ref:SeqRec/models/generative/Qwen3Multi/model.py:744-880 (Qwen3MultiModel.forward),
  :186-247 (decoder layer), :75-150 (attention), :904-922,928-1013 (head + temperature loss),
  ref:SeqRec/models/generative/Qwen3Moe/FFN.py:53-72 (position-routed SwiGLU experts),
and the HF-Trainer update of ref:SeqRec/tasks/train_SMB_decoder.py:396-444.

Data layout in HBM (T = B*S tokens, row t = b*S + s, everything fp32):
  residual stream   [T,H] snapshots per sub-block (needed by the RMSNorm backward)
  q|k|v             [T,(nq+2nkv)*64] fused projection output, v consumed in place by attention
  q_rot, k_rot      [T,nq*64], [T,nkv*64] after per-head RMSNorm + RoPE
  FFN tensors       expert-sorted order [T,*] (slot = expert-major, token order inside an expert),
                    so the three expert GEMMs are dense grouped GEMMs with device-side offsets
  logits            [T, round_up(V,32)]
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .config import Qwen3MultiConfig

IGNORE_INDEX = -100


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class ParamLayout:
    """Flat fp32 layout of all parameters under the reference's state-dict names.

    Decayed parameters first, RMSNorm weights (no weight decay under HF Trainer) last, so the
    fused AdamW kernel needs a single boundary.  q/k/v projection weights of one attention are
    adjacent (one fused [768,256] GEMM operand); the experts' gate and up weights of a layer are stored expert by expert as
    gate_e | up_e (one [2 I, din] operand per expert: the FUSED gate|up projection, input gradient and weight gradient of the
    grouped GEMMs, constant stride 2 I din), their down weights stacked behind them (stride H I).
    """

    VERSION = 2          # 1 (rounds 1-3): the experts' gate / up / down weights of a layer stacked per kind

    def __init__(self, cfg: Qwen3MultiConfig, version: int = 2):
        H, dh = cfg.hidden_size, cfg.head_dim
        nq, nkv, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
        Eb, NB1, E = cfg.behavior_embedding_dim, cfg.num_behavior + 1, cfg.num_experts
        self.version = version
        decay: List[Tuple[str, tuple]] = [("model.embed_tokens.weight", (cfg.vocab_size, H))]
        nodecay: List[Tuple[str, tuple]] = []
        for l in range(cfg.num_hidden_layers):
            lp = f"model.layers.{l}."
            cross = l in cfg.cross_attention_decoder
            inject = l in cfg.behavior_injection_decoder
            for a in (["self_attn", "cross_attn"] if cross else ["self_attn"]):
                ap = lp + a + "."
                decay += [(ap + "q_proj.weight", (nq * dh, H)), (ap + "k_proj.weight", (nkv * dh, H)),
                          (ap + "v_proj.weight", (nkv * dh, H)), (ap + "o_proj.weight", (H, nq * dh))]
                nodecay += [(ap + "q_norm.weight", (dh,)), (ap + "k_norm.weight", (dh,))]
                if a == "cross_attn":
                    decay += [(ap + "gating.weight", (H, H)),
                              (ap + "q_behavior_embedding.weight", (NB1, nq * Eb)),
                              (ap + "k_behavior_embedding.weight", (NB1, nkv * Eb)),
                              (ap + "v_behavior_embedding.weight", (NB1, nkv * Eb))]
            din = H + (Eb if inject else 0)
            if version == 1:
                for kind, shp in (("gate_proj", (I, din)), ("up_proj", (I, din)), ("down_proj", (H, I))):
                    for e in range(E):
                        decay.append((f"{lp}mlp.experts.expert_{e}.{kind}.weight", shp))
            else:
                for e in range(E):
                    decay.append((f"{lp}mlp.experts.expert_{e}.gate_proj.weight", (I, din)))
                    decay.append((f"{lp}mlp.experts.expert_{e}.up_proj.weight", (I, din)))
                for e in range(E):
                    decay.append((f"{lp}mlp.experts.expert_{e}.down_proj.weight", (H, I)))
            if inject:
                decay.append((lp + "mlp.behavior_embedding.weight", (NB1, Eb)))
            nodecay.append((lp + "input_layernorm.weight", (H,)))
            if cross:
                nodecay.append((lp + "post_self_attention_layernorm.weight", (H,)))
            nodecay.append((lp + "post_cross_attention_layernorm.weight", (H,)))
        nodecay.append(("model.norm.weight", (H,)))
        self.entries: Dict[str, Tuple[int, tuple]] = {}
        off = 0
        for name, shp in decay:
            n = math.prod(shp)
            assert n % 4 == 0, f"{name}: size {n} is not a multiple of 4"
            self.entries[name] = (off, shp)
            off += n
        self.n_decay = off
        for name, shp in nodecay:
            n = math.prod(shp)
            assert n % 4 == 0
            self.entries[name] = (off, shp)
            off += n
        self.numel = off

    def views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {k: flat[o:o + math.prod(s)].view(s) for k, (o, s) in self.entries.items()}

    def adopt(self, flat: torch.Tensor, cfg: Qwen3MultiConfig, version) -> torch.Tensor:
        """A flat buffer (optimizer moments of a checkpoint) written under layout `version` (None: a file from before the
        layouts were numbered = 1) re-ordered to this layout, parameter by parameter; the same tensor if nothing moved."""
        version = 1 if version is None else int(version)
        if version > ParamLayout.VERSION or version < 1:
            raise ValueError(f"flat optimizer state was written under parameter layout {version}; this build knows layouts 1.."
                             f"{ParamLayout.VERSION} (a checkpoint of a newer gamer_amd?)")
        if version == self.version:
            return flat
        if flat.numel() != self.numel:
            raise ValueError(f"flat optimizer state has {flat.numel()} elements, this model {self.numel}")
        old = ParamLayout(cfg, version=version).views(flat)
        out = torch.empty_like(flat)
        for k, v in self.views(out).items():
            v.copy_(old[k])
        return out

    def span(self, flat: torch.Tensor, first: str, rows: int, cols: int) -> torch.Tensor:
        """[rows, cols] view that starts at parameter `first` and runs over its successors."""
        o = self.entries[first][0]
        return flat[o:o + rows * cols].view(rows, cols)


class _LayerW:
    """Fused weight views of one decoder layer over a flat buffer (parameters or gradients)."""

    def __init__(self, cfg, layout: ParamLayout, flat: torch.Tensor, l: int):
        H, dh = cfg.hidden_size, cfg.head_dim
        nq, nkv, I, E = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size, cfg.num_experts
        QKV = (nq + 2 * nkv) * dh
        v = layout.views(flat)
        lp = f"model.layers.{l}."
        self.cross = l in cfg.cross_attention_decoder
        self.inject = l in cfg.behavior_injection_decoder
        self.din = H + (cfg.behavior_embedding_dim if self.inject else 0)

        def attn(prefix):
            d = dict(qkv=layout.span(flat, prefix + "q_proj.weight", QKV, H), o=v[prefix + "o_proj.weight"],
                     qn=v[prefix + "q_norm.weight"], kn=v[prefix + "k_norm.weight"])
            if prefix.endswith("cross_attn."):
                d.update(gate=v[prefix + "gating.weight"], bq=v[prefix + "q_behavior_embedding.weight"],
                         bk=v[prefix + "k_behavior_embedding.weight"], bv=v[prefix + "v_behavior_embedding.weight"])
            return d
        self.self_attn = attn(lp + "self_attn.")
        self.cross_attn = attn(lp + "cross_attn.") if self.cross else None
        self.ln1 = v[lp + "input_layernorm.weight"]
        self.ln2 = v[lp + "post_self_attention_layernorm.weight"] if self.cross else None
        self.ln3 = v[lp + "post_cross_attention_layernorm.weight"]
        # rows [e 2I, e 2I + I) = gate_proj of expert e, [e 2I + I, (e + 1) 2I) = its up_proj
        self.gu = layout.span(flat, f"{lp}mlp.experts.expert_0.gate_proj.weight", E * 2 * I, self.din)
        self.I = I
        self.down = layout.span(flat, f"{lp}mlp.experts.expert_0.down_proj.weight", E * H, I)
        self.beh = v[lp + "mlp.behavior_embedding.weight"] if self.inject else None

    def gate_e(self, e: int) -> torch.Tensor:
        return self.gu[e * 2 * self.I:e * 2 * self.I + self.I]

    def up_e(self, e: int) -> torch.Tensor:
        return self.gu[e * 2 * self.I + self.I:(e + 1) * 2 * self.I]


class Bf16Shadow:
    """bf16 operand copies of the fp32 master parameters for the AMP variant (the reference's autocast casts the weight
    of every F.linear to bf16 on the fly; here the cast happens once per optimizer step):

      flat16   same offsets as the fp32 flat buffer (k-contiguous operands of the forward and weight-gradient GEMMs)
      flatT    the TRANSPOSED matrices, so that the input-gradient GEMMs dX = dY W are k-contiguous on both sides as
               well (gamer_gemm_bf16); the tied head's transpose is [H, round_up(V, 64)] with zero padding (its
               contraction runs over the padded vocabulary).
    One gamer_cast_params_bf16 launch refreshes both (98 MB read, 98 MB written)."""

    def __init__(self, cfg: Qwen3MultiConfig, layout: ParamLayout, flat_p: torch.Tensor):
        dev = flat_p.device
        H, dh = cfg.hidden_size, cfg.head_dim
        nq, nkv, I, E = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size, cfg.num_experts
        QKV = (nq + 2 * nkv) * dh
        self.layout, self.flat_p = layout, flat_p
        self.flat16 = torch.zeros(layout.numel, dtype=torch.bfloat16, device=dev)
        self.ldv = _round_up(cfg.vocab_size, 64)
        entries = []          # (src offset, rows, cols, ldt, key of the transposed view)
        toff = 0
        self.t_views: Dict[str, Tuple[int, tuple]] = {}

        def add(first: str, rows: int, cols: int, ldt: Optional[int] = None, tkey: Optional[str] = None):
            nonlocal toff
            src = layout.entries[first][0]
            ldt = ldt or rows
            entries.append((src, rows, cols, ldt, toff))
            self.t_views[tkey or first] = (toff, (cols, ldt))
            toff += _round_up(cols * ldt, 8)

        add("model.embed_tokens.weight", cfg.vocab_size, H, ldt=self.ldv)
        for l in range(cfg.num_hidden_layers):
            lp = f"model.layers.{l}."
            cross = l in cfg.cross_attention_decoder
            din = H + (cfg.behavior_embedding_dim if l in cfg.behavior_injection_decoder else 0)
            for a in (["self_attn", "cross_attn"] if cross else ["self_attn"]):
                add(lp + a + ".q_proj.weight", QKV, H, tkey=lp + a + ".qkv")
                add(lp + a + ".o_proj.weight", H, nq * dh)
                if a == "cross_attn":
                    add(lp + a + ".gating.weight", H, H)
            for e in range(E):              # gate_e | up_e are adjacent: one [2 I, din] matrix per expert, transposed [din, 2 I]
                add(f"{lp}mlp.experts.expert_{e}.gate_proj.weight", 2 * I, din, tkey=f"{lp}mlp.experts.expert_{e}.gu")
            for e in range(E):
                add(f"{lp}mlp.experts.expert_{e}.down_proj.weight", H, I)
        self.flatT = torch.zeros(toff, dtype=torch.bfloat16, device=dev)
        tab, tile0 = [], 0
        for src, rows, cols, ldt, dst_t in entries:
            tab += [src, src, dst_t, rows | (cols << 32), ldt | (tile0 << 32)]
            tile0 += ((rows + 31) // 32) * ((cols + 31) // 32)
        self.n_entries, self.n_tiles = len(entries), tile0
        self.table = torch.tensor(tab, dtype=torch.int64, device=dev)
        self.params16 = layout.views(self.flat16)

    def refresh(self):
        ops.cast_params_bf16(self.flat_p, self.flat16, self.flatT, self.table, self.n_entries, self.n_tiles)

    def t(self, key: str) -> torch.Tensor:
        """[cols, ldt] view of the transposed copy of matrix `key`."""
        off, shp = self.t_views[key]
        return self.flatT[off:off + shp[0] * shp[1]].view(shp)


class _LayerWT:
    """Transposed bf16 weight views of one decoder layer (dgrad operands of the AMP variant)."""

    def __init__(self, cfg, sh: Bf16Shadow, l: int):
        lp = f"model.layers.{l}."
        cross = l in cfg.cross_attention_decoder

        def attn(a):
            d = dict(qkv=sh.t(lp + a + ".qkv"), o=sh.t(lp + a + ".o_proj.weight"))
            if a == "cross_attn":
                d["gate"] = sh.t(lp + a + ".gating.weight")
            return d
        self.self_attn = attn("self_attn")
        self.cross_attn = attn("cross_attn") if cross else None
        # expert e's transposed matrix sits e * (cols * rows) behind expert 0's (consecutive add() calls, sizes % 8 == 0)
        self.gu = sh.t(f"{lp}mlp.experts.expert_0.gu")
        self.down = sh.t(f"{lp}mlp.experts.expert_0.down_proj.weight")


class _Workspace:
    """All activation / scratch buffers of the train step (``train=True``) or of a scoring forward.

    Storage is grow-only and shape independent: every buffer is a flat allocation that ``bind(B, S)`` views at the
    shape of the current batch, so collated batches of varying length (the reference's collator pads to the longest
    sample of each batch, ref:SeqRec/datasets/collator.py:59-67) reuse one allocation instead of freeing and
    re-creating tens of GB per micro-batch; ``Engine.reserve`` sizes it once for the largest batch up front.
    ``act`` is the dtype of the activations the GEMMs read and write: fp32, or bf16 for the AMP variant (the
    residual stream, the normalisation statistics and everything the optimizer touches stay fp32)."""

    def __init__(self, cfg: Qwen3MultiConfig, device, train: bool, act: torch.dtype = torch.float32, spill: bool = True):
        self.cfg, self.device, self.train, self.act, self.spill = cfg, device, train, act, spill
        self._store: Dict[str, torch.Tensor] = {}
        self.B = self.S = self.T = 0
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=device)
        self.count = torch.zeros(1, dtype=torch.float32, device=device)
        self.session: Optional[dict] = None        # span_self / span_cross / pos_ids of the session variant

    def _buf(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        n = math.prod(shape)
        t = self._store.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            self._store[name] = None               # release the old block before asking for the larger one
            t = torch.empty(n, dtype=dtype, device=self.device)
            self._store[name] = t
        return t[:n].view(shape)

    def allocated_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._store.values() if t is not None)

    def bind(self, B: int, S: int):
        if (B, S) == (self.B, self.S):
            return self
        cfg, train, act = self.cfg, self.train, self.act
        f32, i32 = torch.float32, torch.int32
        self.B, self.S, self.T = B, S, B * S
        T, H = self.T, cfg.hidden_size
        nq, nkv, dh, I = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size
        NQ, NKV = nq * dh, nkv * dh
        QKV = NQ + 2 * NKV
        L, E = cfg.num_hidden_layers, cfg.num_experts
        din_max = H + cfg.behavior_embedding_dim
        n_t32 = (S + 31) // 32
        buf = self._buf
        self.router = {k: buf("r_" + k, (B, S), i32) for k in
                       ("expert", "beh_idx", "act_idx", "kl_self", "kl_cross", "ql_cross", "empty_self", "empty_cross")}
        self.router["tile_empty_self"] = buf("r_tile_empty_self", (B, n_t32), i32)
        self.router["tile_empty_cross"] = buf("r_tile_empty_cross", (B, n_t32), i32)
        if "r_bad_token" not in self._store:
            self._store["r_bad_token"] = torch.zeros(1, dtype=i32, device=self.device)
            self._store["r_bad_label"] = torch.zeros(1, dtype=i32, device=self.device)
        self.router["bad_token"] = self._store["r_bad_token"]
        self.bad_label = self._store["r_bad_label"]
        if self.session is not None:
            self.session = None                    # rebuilt at the new shape by the next session forward
        self.perm = buf("perm", (T,), i32)
        self.slot = buf("slot", (T,), i32)
        # query-row order of the cross attention (normal rows first, empty rows behind)
        self.cross_order = (buf("co_perm", (B, S), i32), buf("co_kind", (B, n_t32), i32), buf("co_maxpos", (B, n_t32), i32))
        self.offsets = buf("offsets", (E + 1,), i32)
        self.work = buf("work", ((B + 1) * E,), i32)
        # logits [T, ldl]: fp32, or bf16 as the reference's autocast lm_head returns them; the bf16 row is padded to a
        # multiple of 64 columns (the head's dgrad contracts over the padded vocabulary) and the padding stays zero
        self.ldl = _round_up(cfg.vocab_size, 32 if act == f32 else 64)
        self.logits = buf("logits", (T, self.ldl), act)
        if act != f32:
            self.logits[:, cfg.vocab_size:].zero_()
        self.lse_ce = buf("lse_ce", (T,), f32)
        self.row_loss = buf("row_loss", (T,), f32)
        self.xn = buf("xn", (T, H), act)
        # residual stream snapshots: x[l][0] layer input, [1] after self-attn, [2] after cross-attn
        self.x: List[List[torch.Tensor]] = []
        self.layers: List[dict] = []
        keep = train            # eval keeps one set of buffers and reuses it for every layer
        shared: Optional[dict] = None
        for l in range(L):
            cross = l in cfg.cross_attention_decoder
            tag = f"l{l}_" if keep else "l_"
            if keep or shared is None:
                d = dict(
                    h1=buf(tag + "h1", (T, H), act), qkv=buf(tag + "qkv", (T, QKV), act), q=buf(tag + "q", (T, NQ), act),
                    k=buf(tag + "k", (T, NKV), act), ao=buf(tag + "ao", (T, NQ), act), lse=buf(tag + "lse", (B, nq, S), f32),
                    hin=buf(tag + "hin", (T, din_max), act), gu=buf(tag + "gu", (T, 2 * I), act),
                    hm=buf(tag + "hm", (T, I), act))
                if cross or not keep:
                    d.update(h2=buf(tag + "h2", (T, H), act), qkv_c=buf(tag + "qkv_c", (T, QKV), act),
                             q_c=buf(tag + "q_c", (T, NQ), act), k_c=buf(tag + "k_c", (T, NKV), act),
                             ao_c=buf(tag + "ao_c", (T, NQ), act), lse_c=buf(tag + "lse_c", (B, nq, S), f32),
                             op_c=buf(tag + "op_c", (T, H), act), gate_c=buf(tag + "gate_c", (T, H), act))
                shared = d
            self.layers.append(shared if not keep else d)
            if keep or l == 0:
                xs = [buf(tag + "x0", (T, H), f32), buf(tag + "x1", (T, H), f32)]
                xs.append(buf(tag + "x2", (T, H), f32) if (cross or not keep) else None)
                self.x.append(xs)
            else:
                self.x.append(self.x[0])
        self.x_final = buf("x_final", (T, H), f32) if keep else self.x[0][0]
        # scratch shared by all layers
        self.tmpH = [buf(f"tmpH{i}", (T, H), act) for i in range(4)]
        if train:
            self.dx = buf("dx", (T, H), f32)
            self.dhm = buf("dhm", (T, I), act)
            self.dhin = buf("dhin", (T, din_max), act)
            self.dqkv = buf("dqkv", (T, QKV), act)
            self.dq = buf("dq", (T, NQ), act)
            self.dk = buf("dk", (T, NKV), act)
            self.dao = buf("dao", (T, NQ), act)
            self.delta = buf("delta", (B, nq, S), f32)
            # one [2048, H] table of weight-gradient partial sums per RMSNorm backward of a pass (reduced together at its end),
            # plus one scratch table for the ordered behaviour-table gradient
            self.norm_partial = buf("norm_partial", (3 * L + 2, 2048, H), f32)
            self.qk_partial = buf("qk_partial", (ops.qknorm_partial_numel(cfg.num_behavior + 1),), f32)
            # dS spill of the attention backward (5 matmuls per tile pair instead of 7; gamer_attn_bwd):
            # 6.4 GB at B = 1024, shared by all layers.  GAMER_ATTN_SPILL=0 keeps the two recompute kernels.
            import os
            self.ds_work = (buf("ds_work", (ops.attn_ds_work_numel(B, S, nq),), f32)
                            if (os.environ.get("GAMER_ATTN_SPILL", "1") != "0" and act == torch.float32 and self.spill)
                            else None)
        return self


class Engine:
    """Owns the flat parameter/gradient/optimizer buffers and runs forward / backward / update."""

    N_NORM_PARTIAL = 512
    N_SUMSQ_PARTIAL = 1024

    def __init__(self, cfg: Qwen3MultiConfig, device="cuda", temperature: float = 1.0, variant: str = "multi",
                 dtype: str = "f32", matmul: Optional[str] = None, share_buffers_of: Optional["Engine"] = None,
                 deterministic: Optional[bool] = None):
        """``variant``: "multi" = Qwen3Multi (train_SMB_decoder.py:362-364), "session" = Qwen3SessionMulti
        (train_SMB_decoder.py:365-367): the same parameters and layers with session-wise attention masks and
        RoPE positions taken from ``extended_session_ids``.
        ``dtype``: "f32" (default) or "bf16" = the reference's ``--bf16`` run (train_SMB_decoder.py:114-118, 407-408:
        HF Trainer autocast): bf16 matrix operands and activations, fp32 accumulation, fp32 residual stream,
        normalisation statistics, loss, master weights, gradients and optimizer state.
        ``matmul`` (dtype "f32" only): how the nn.Linear sites multiply fp32 operands - "f32" = v_mfma_f32_32x32x2_f32,
        "split6" / "split9" = gamer_gemm_f32_split: both operands cut exactly into three bf16 pieces, 6 or 9 piece
        products on the 16x wider bf16 pipe, fp32 accumulation (include/gamer_hip.h; error vs fp64 measured equal to
        the fp32 MFMA's, tools/split_error.py).  The attention products take the same form (gamer_attn_fwd_split /
        gamer_attn_bwd_split, csrc/attention_split.hip; ``self.split_attention = False`` keeps them on the fp32 MFMA).
        "split3" (the default) = the Linear sites as THREE fp16 piece products of a two-way cut of both operands, each
        operand tensor scaled by a power of two from its largest magnitude (gamer_absmax_f32; docs/DESIGN_rounds1-4.md section 17) - the
        same error against fp64 at half the matrix instructions; the attention products take the same three-product form
        (``self.h2_attention = False`` keeps them in the six-product bf16 form).
        ``deterministic`` (fp32 forms; None = GAMER_DETERMINISTIC=1): EVERY reduction of the step in a fixed order, so that two
        runs from the same state give the same bits.  The weight gradients (two-pass chunk reduce), the behaviour-table
        gradients and the embedding gradient (stable counting sort + sums in token order, gamer_embedding_bwd_ordered) are
        ordered by default in the fp32 forms; what this flag adds is that it FORCES those forms whatever the environment says
        (GAMER_WGRAD_TWO_PASS / GAMER_EMBEDDING_ATOMICS), selects the ordered two-pass weight gradients of the bf16 step
        (opt-in there: +0.6 ms), and raises instead of falling back when an ordered form cannot be used (vocabulary above
        8191 rows: the ordered embedding gradient's row index is 13 bits).
        ``share_buffers_of``: another engine of the same configuration whose flat parameter and gradient buffers this one
        uses instead of allocating its own - the nn.Module keeps one set of fp32 masters and runs them through the fp32 or the
        bf16 step depending on the caller's autocast state (gamer_amd/modeling.py)."""
        cfg.validate()
        if matmul is None:
            # default of the fp32 path: products on the 16-bit matrix pipe from piece cuts (docs/DESIGN_rounds1-4.md sections 13, 15, 17);
            # "f32" = fp32 MFMA
            matmul = "split3" if dtype == "f32" else "f32"
        if matmul not in ops.MATMUL_MODES:
            raise ValueError(f"unknown matmul {matmul!r} ({sorted(ops.MATMUL_MODES)})")
        if dtype != "f32" and matmul != "f32":
            raise ValueError("matmul='split3'/'split6'/'split9' is a form of the fp32 path; dtype='bf16' has its own GEMM")
        self.matmul = matmul
        self.deterministic = (os.environ.get("GAMER_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        # q / k RMSNorm + RoPE in the q|k|v projection's epilogue (gamer_gemm_desc.qk_*): built, parity-tested and measured
        # at batch 1024 - the 12 projections got 6.8 ms slower per step (their tiles now also store q_rot / k_rot, and
        # on gfx950 epilogue work is matrix-pipe time), the removed kernel had cost 8.1 ms: 373.9-375.4 against 373.4-377 ms
        # per step, inside the box-to-box noise.  Off by default; Engine.fuse_qkv = True turns it on.
        self.fuse_qkv = False
        # the same epilogue in the bf16 GEMM (gamer_gemm_bf16_desc.qk_*; round 3): the twelve projections got 4.5 ms slower per
        # step at batch 1024 (9.4 -> 13.9 ms), the removed kernel had cost 4.2 ms: 108.4 vs 108.7 ms, 16.7 vs 16.6 ms at batch 128
        # - neutral again.  Off by default; GAMER_FUSE_QKV_BF16=1 or Engine.fuse_qkv_bf16 = True turns it on.
        self.fuse_qkv_bf16 = os.environ.get("GAMER_FUSE_QKV_BF16", "0") == "1"
        # attention products of the split forms on the bf16 pipe too (csrc/attention_split.hip); False keeps fp32-MFMA attention
        self.split_attention = True
        if variant not in ("multi", "session"):
            raise ValueError(f"unknown variant {variant!r}")
        if dtype not in ("f32", "bf16"):
            raise ValueError(f"unknown dtype {dtype!r} (f32 or bf16; the reference's --fp16 is not built)")
        self.variant = variant
        self.dtype = dtype
        self.act_dtype = torch.float32 if dtype == "f32" else torch.bfloat16
        if not torch.cuda.is_available():
            raise RuntimeError("gamer_amd.Engine needs a HIP device (there is no CPU fallback)")
        from . import _lib
        _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self.temperature = float(temperature)
        self.layout = ParamLayout(cfg)
        n = self.layout.numel
        if share_buffers_of is not None:
            if share_buffers_of.layout.numel != n or share_buffers_of.device != self.device:
                raise ValueError("share_buffers_of: the other engine has another parameter layout or device")
            self.flat_p, self.flat_g = share_buffers_of.flat_p, share_buffers_of.flat_g
        else:
            self.flat_p = torch.zeros(n, dtype=torch.float32, device=self.device)
            self.flat_g = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.flat_m: Optional[torch.Tensor] = None
        self.flat_v: Optional[torch.Tensor] = None
        self.params = self.layout.views(self.flat_p)
        self.grads = self.layout.views(self.flat_g)
        L = cfg.num_hidden_layers
        self.W = [_LayerW(cfg, self.layout, self.flat_p, l) for l in range(L)]
        self.G = [_LayerW(cfg, self.layout, self.flat_g, l) for l in range(L)]
        # matrices the GEMMs read: the fp32 masters themselves, or their bf16 copies (+ transposed copies for dgrad)
        self.shadow: Optional[Bf16Shadow] = None
        self.Wm, self.WT = self.W, None
        if dtype == "bf16":
            self.shadow = Bf16Shadow(cfg, self.layout, self.flat_p)
            self.Wm = [_LayerW(cfg, self.layout, self.shadow.flat16, l) for l in range(L)]
            self.WT = [_LayerWT(cfg, self.shadow, l) for l in range(L)]
        # split forms, opt-in (GAMER_SPLIT_PLANES=1): the fp32 masters cut once per step into three bf16 planes - the B operand
        # of every forward / input-gradient GEMM is a weight, and without this every row tile of the activations cuts its weight
        # tile again (half of a K-step's cut instructions).  Bit-identical results (tests/test_ops_gpu.py); measured at batch
        # 1024: 320.98 vs 320.55 ms per step, forward GEMMs 36.2 vs 35.1 ms, input-gradient GEMMs 43.6 vs 44.0 - the cut
        # arithmetic is not what bounds the kernel (docs/DESIGN_rounds1-4.md section 13), so it stays off.
        self.weight_planes: Optional[torch.Tensor] = None
        if dtype == "f32" and matmul != "f32" and os.environ.get("GAMER_SPLIT_PLANES", "0") == "1":
            self.weight_planes = torch.zeros(3, _round_up(n, 4), dtype=torch.bfloat16, device=self.device)
        # matmul="split3": the per-tensor maxima its GEMMs scale their operands by; the parameters keep theirs for a whole pass
        # matmul="split3": the attention products in the three-product fp16 form as well (gamer_attn_split_amax); False keeps
        # them in the six-product bf16 form
        self.h2_attention = os.environ.get("GAMER_H2_ATTENTION", "1") != "0"
        # max |v| / max |dO| of the three-product attention out of the producing GEMMs' epilogues (gamer_gemm_desc.amax_c) instead of
        # gamer_absmax_f32 passes; GAMER_GEMM_CAMAX=0 keeps the passes (A/B)
        self.gemm_c_amax = os.environ.get("GAMER_GEMM_CAMAX", "1") != "0"
        self._amax = None
        if dtype == "f32" and matmul == "split3":
            self._amax = ops.amax_reuse()
            self._amax.stable_range(self.flat_p.data_ptr(), self.flat_p.numel() * 4)
            if os.environ.get("GAMER_SPLIT3_PLANES", "1") != "0":
                # the parameters' fp16 pieces, packed at their values' offsets and rebuilt at the start of every pass (98 MB read,
                # 98 MB written): the B operand of the forward and input-gradient GEMMs is then staged without its cut
                self._amax.planes = torch.zeros(_round_up(n, 4), dtype=torch.float32, device=self.device)
                if cfg.hidden_size == 256 and os.environ.get("GAMER_GEMM_OSF", "0") == "1":
                    # GAMER_GEMM_OSF=1 (A/B): ... and the pieces of the TRANSPOSES of the weights with 256 output rows and a longer
                    # contraction (o_proj, the experts' down projection): their forward then runs on the output-stationary kernel
                    # (gamer_gemm_desc.b_planes_t).  MEASURED slower than the 128 x 128 kernel's residual epilogue (same box: 9.6
                    # against 9.4 ms per step for the sixteen launches, 210.8 against 210.2 ms per step) - off by default.
                    self._amax.planes_t = torch.zeros(_round_up(n, 4), dtype=torch.float32, device=self.device)
        self.lut = cfg.behavior_lut().to(self.device)
        self._rope: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._ws: Dict[bool, _Workspace] = {}
        self.ws: Optional[_Workspace] = None
        self.opt_step = 0
        self.dropout_step = 0
        self.base_seed = 0x5EED
        self.sumsq_partial = torch.empty(self.N_SUMSQ_PARTIAL, dtype=torch.float32, device=self.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._norm_out_table = None                 # (addresses, device array) of the RMSNorm weight gradients in backward order
        self.fuse_swiglu_bwd = os.environ.get("GAMER_FUSE_SWIGLU_BWD", "1") != "0"     # (A/B switch; split3 only)
        # fp32 forms: the injecting layers' gate|up projection runs on the 256 hidden columns only; the behaviour-embedding columns'
        # share (NB + 1 different rows per expert) is a table added where gate|up are consumed (csrc/inject.hip).  Rows are then
        # sorted by (expert, behaviour).  GAMER_SPLIT_INJECT=0: the reference's concatenated [T, 320] input (A/B runs).
        nb1 = cfg.num_behavior + 1
        self.split_inject = bool(self.dtype != "bf16" and os.environ.get("GAMER_SPLIT_INJECT", "1") != "0" and
                                 cfg.behavior_injection_decoder and cfg.num_experts * nb1 <= 64 and nb1 <= 16 and
                                 cfg.behavior_embedding_dim % 4 == 0 and cfg.behavior_embedding_dim <= 256)
        self.ordered_embedding_grad = os.environ.get("GAMER_EMBEDDING_ATOMICS", "0") == "0"   # (1: the float-atomics scatter)
        self._check_deterministic_embedding()
        self._saved = None

    def _check_deterministic_embedding(self):
        V = self.cfg.vocab_size
        if self.ordered_embedding_grad and V > 8191 and self.deterministic:
            raise RuntimeError(f"Engine(deterministic=True): the ordered embedding gradient takes at most 8191 vocabulary rows (V = {V}); "
                               "the float-atomics scatter it would fall back to is not reproducible")

    # ------------------------------------------------------------------------------------------
    def init_weights(self, seed: int = 0):
        """normal(0, initializer_range) for matrices/embeddings, ones for RMSNorm weights, zero padding
        row (HF _init_weights semantics; the reference's resize_token_embeddings mean-resizing is not
        reproduced)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        std = self.cfg.initializer_range
        for name, (off, shp) in self.layout.entries.items():
            n = math.prod(shp)
            if len(shp) == 1:
                self.flat_p[off:off + n].fill_(1.0)
            else:
                self.flat_p[off:off + n].copy_(torch.randn(n, generator=g) * std)
        self.params["model.embed_tokens.weight"][self.cfg.pad_token_id].zero_()

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        missing = [k for k in self.layout.entries if k not in sd]
        if missing:
            raise KeyError(f"missing parameters: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        for k, p in self.params.items():
            if tuple(sd[k].shape) != tuple(p.shape):
                raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {tuple(p.shape)}")
            p.copy_(sd[k].to(torch.float32))

    def rope(self, S: int):
        if S not in self._rope:
            # Qwen3RotaryEmbedding.forward with position_ids = arange(S), fp32 on the host (bit-identical
            # to the table the reference builds on CPU).
            dh, theta = self.cfg.head_dim, float(self.cfg.rope_theta)
            inv_freq = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.int64).to(torch.float32) / dh))
            freqs = torch.arange(S, dtype=torch.float32)[:, None] * inv_freq[None, :]
            emb = torch.cat((freqs, freqs), dim=-1)
            self._rope[S] = (emb.cos().to(self.device).contiguous(), emb.sin().to(self.device).contiguous())
        return self._rope[S]

    def workspace(self, B: int, S: int, train: bool) -> _Workspace:
        """The train and the scoring workspace live side by side (an epoch's evaluation does not evict the training
        buffers); each is bound to the shape of the current batch without reallocating (grow-only storage)."""
        if train not in self._ws:
            # the 6.4 GB dS-spill scratch is the fp32-MFMA attention backward's; the split form (recompute) does not use
            # it - unless the session variant sends its span calls to the fp32-MFMA kernels
            # (the three-product form takes the session variant's key spans itself)
            spill = not (self.split_attention and self.matmul != "f32" and
                         (self.variant == "multi" or (self.matmul == "split3" and self.h2_attention)))
            self._ws[train] = _Workspace(self.cfg, self.device, train, self.act_dtype, spill=spill)
        return self._ws[train].bind(B, S)

    def reserve(self, B: int, S: int, train: bool = True):
        """Size the workspace once for the largest batch that will be seen (e.g. S = (max_his_len + 1) * num_positions),
        so that shorter collated batches never allocate."""
        self.workspace(B, S, train)

    def _seed(self, layer: int, site: int) -> int:
        return ((self.base_seed & 0xFFFF) << 48) | ((self.dropout_step & 0xFFFFFFFF) << 16) | (layer << 4) | site

    # ------------------------------------------------------------------------------------------
    def _planes(self):
        wp = self.weight_planes
        return None if wp is None else (self.flat_p.data_ptr(), self.flat_p.numel() * 4, wp.data_ptr(), wp.stride(0))

    @ops.scoped_f32_matmul(lambda self, *a: self.matmul, lambda self, *a: self._planes())
    @ops.scoped_amax(lambda self, *a: self._amax)
    def forward(self, input_ids, attention_mask=None, actions=None, labels=None, num_items_in_batch=None,
                train: bool = False, dropout: Optional[bool] = None, act_zero_col: Optional[int] = None,
                uniform_len: int = 0, kv_sink=None, kv_dest=None, session_ids=None, extended_session_ids=None,
                last_row_logits: bool = False, hidden_sink: Optional[list] = None):
        """Returns (loss or None, logits view [B,S,V]).  With labels the logits are divided by the
        temperature in place, as the reference does (model.py:913).  The view aliases a workspace
        buffer: backward() overwrites it with d(logits), the next forward() with new logits.  ``train`` keeps every activation
        needed by backward(); ``dropout`` defaults to ``train``.

        ``session_ids`` / ``extended_session_ids`` [B,S] are required by (and only used in) the "session" variant
        (Qwen3SessionMulti: session-wise masks, RoPE positions = extended_session_ids).

        Evaluation only (gamer_amd/decode.py re-runs the whole sequence every generation step): ``act_zero_col``
        is the column of the prompt's final behaviour token, which the reference's router gives action index 0
        when the prompt has 5n+1 tokens (router.py:160-163) and which stays that way in its K/V cache;
        ``uniform_len`` is the prompt length (see gamer_attn_fwd in include/gamer_hip.h); ``kv_dest(layer, kind)`` -> (q|k|v buffer [T, QKV],
        rotated-key buffer [T, NKV]) makes the layer write them there; ``kv_sink(layer, kind, k, v)``
        receives the keys (after k-norm + RoPE) and values of every attention ("self" / "cross") - the K/V cache of
        gamer_amd.decode.DecodeSession; ``last_row_logits``: the prompt pass of a generation needs the logits of the
        last position only (HF's ``logits_to_keep``): final norm and head run on B rows instead of B*S and the
        returned logits are [B, 1, V].

        ``hidden_sink``: a list that receives copies [B, S, H] of the residual stream at the entry of every decoder layer and of
        the final norm's output - HF's ``output_hidden_states`` tuple (model.py:822-873).  fp32 in both dtypes (the residual
        stream is kept in fp32), except the last entry, which has the activation dtype."""
        cfg = self.cfg
        B, S = input_ids.shape
        if train and S % cfg.num_positions != 0:
            raise ValueError(f"sequence length {S} is not a multiple of num_positions={cfg.num_positions} "
                             "(the router assumes item-aligned sequences, router.py:78-81)")
        if train and (act_zero_col is not None or uniform_len not in (0, S)):
            raise ValueError("act_zero_col / uniform_len are evaluation-only options")
        if kv_dest is not None and train:
            raise ValueError("kv_dest is an evaluation-only option (the backward reads the workspace's q|k|v and keys)")
        if last_row_logits and (train or labels is not None):
            raise ValueError("last_row_logits is an evaluation-only option")
        if hidden_sink is not None and last_row_logits:
            raise ValueError("hidden_sink needs the full-sequence forward")
        bf16 = self.dtype == "bf16"
        if bf16:
            if last_row_logits or uniform_len not in (0, S):
                raise NotImplementedError("generation (cached decode / re-run scoring) is built for dtype='f32' only")
            self.shadow.refresh()                    # the masters may have been updated by any optimizer since the last call
        if self.weight_planes is not None:
            ops.split3_planes(self.flat_p, self.weight_planes)     # likewise: 98 MB read, 147 MB written, once per forward
        T, H = B * S, cfg.hidden_size
        nq, nkv, dh, I, E = (cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size,
                             cfg.num_experts)
        NQ, NKV = nq * dh, nkv * dh
        QKV = NQ + 2 * NKV
        eps = float(cfg.rms_norm_eps)
        use_drop = train if dropout is None else dropout
        p_res = float(cfg.dropout_rate) if use_drop else 0.0
        p_att = float(cfg.attention_dropout) if use_drop else 0.0
        if use_drop:
            self.dropout_step += 1
        ws = self.workspace(B, S, train)
        self.ws = ws
        if self._amax is not None:
            # matmul="split3": maxima of the tensors that do not change between their first GEMM of this pass and the backward
            # (saved GEMM inputs, the parameters) are measured once; everything else per GEMM (ops.amax_reuse)
            self._amax.reset()
            if train:       # (the evaluation workspace shares one set of buffers between the layers: nothing is stable there)
                self._amax.stable(ws.xn, *[A[k] for A in ws.layers for k in ("h1", "h2", "ao", "ao_c", "hin", "hm") if k in A])
                if self.h2_attention:       # the attention operands saved for the backward: q, k (after norm + RoPE) and the v columns
                    self._amax.stable(*[A[k] for A in ws.layers for k in ("q", "k", "q_c", "k_c") if k in A],
                                      *[A[k][:, NQ + NKV:] for A in ws.layers for k in ("qkv", "qkv_c") if k in A])
        ids = input_ids.to(self.device, torch.int64).contiguous()
        am = attention_mask.to(self.device, torch.int64).contiguous() if attention_mask is not None else None
        act = actions.to(self.device, torch.int64).contiguous() if actions is not None else None
        lab = labels.to(self.device, torch.int64).contiguous() if labels is not None else None
        r = ws.router
        r["bad_token"].zero_()
        if lab is not None:
            ws.bad_label.zero_()
            ops.check_labels(lab, cfg.vocab_size, IGNORE_INDEX, ws.bad_label)
        ops.router_fwd(ids, am, act, self.lut, cfg.num_positions, cfg.pad_token_id, cfg.eos_token_id, r)
        if act_zero_col is None and S % cfg.num_positions == 1:
            # the reference's router counts (S + 3) // 5 items (router.py:160-163), so a trailing behaviour token
            # - an evaluation prompt - gets action index 0
            act_zero_col = S - 1
        if act_zero_col is not None:
            r["act_idx"][:, act_zero_col] = 0
        NB1 = cfg.num_behavior + 1
        if self.split_inject:
            # rows sorted by (expert, behaviour): group g = expert * NB1 + behaviour; the experts' segments are every NB1-th offset
            ws.grp_key = ws._buf("grp_key", (B, S), torch.int32)
            ws.grp_offsets = ws._buf("grp_offsets", (E * NB1 + 1,), torch.int32)
            ws.row_group = ws._buf("row_group", (T,), torch.int32)
            torch.add(r["beh_idx"], r["expert"], alpha=NB1, out=ws.grp_key)
            ops.expert_lists(ws.grp_key, E * NB1, ws.perm, ws.slot, ws.grp_offsets, ws._buf("grp_work", ((B + 1) * E * NB1,), torch.int32))
            ws.offsets.copy_(ws.grp_offsets[::NB1])
            torch.index_select(ws.grp_key.view(-1), 0, ws.perm, out=ws.row_group)
            if self._amax is not None:
                self._amax.register(*[self.W[l].gu for l in cfg.behavior_injection_decoder])
        else:
            ops.expert_lists(r["expert"], E, ws.perm, ws.slot, ws.offsets, ws.work)
        if self._amax is not None and self._amax.planes_t is not None:
            for Wl in self.W:
                self._amax.register_transposed(Wl.self_attn["o"], 1, H, NQ)
                self._amax.register_transposed(Wl.down, E, H, I)
                if Wl.cross:
                    self._amax.register_transposed(Wl.cross_attn["o"], 1, H, NQ)
        span_self = span_cross = pos_ids = None
        if self.variant == "session":
            # Qwen3SessionMulti/model.py:784-806: both masks are rebuilt from session_ids on every forward
            if session_ids is None:
                raise ValueError("Session IDs must be provided to generate session-wise causal mask.")
            sid = session_ids.to(self.device, torch.int64).contiguous()
            ext = (extended_session_ids.to(self.device, torch.int64).contiguous()
                   if extended_session_ids is not None else None)
            if ws.session is None:
                i32 = torch.int32
                if "s_violations" not in ws._store:
                    ws._store["s_violations"] = torch.zeros(1, dtype=i32, device=self.device)
                ws.session = {"span_self": ws._buf("s_span_self", (B, S, 4), i32),
                              "span_cross": ws._buf("s_span_cross", (B, S, 4), i32),
                              "pos_ids": ws._buf("s_pos_ids", (B, S), i32), "violations": ws._store["s_violations"]}
            ws.session["violations"].zero_()
            ops.session_spans(sid, ext, am, cfg.num_positions, S, r, ws.session)     # overwrites r["empty_*"]
            span_self, span_cross, pos_ids = (ws.session["span_self"], ws.session["span_cross"],
                                              ws.session["pos_ids"])
        if cfg.cross_attention_decoder:
            # (bf16: the rows without an allowed key - 70 % of them in the cross attention - output 0; sorted behind the
            # others they cost nothing)
            ops.attn_row_order(r["empty_cross"], *ws.cross_order)
        cos, sin = self.rope(S)
        scale = float(dh) ** -0.5
        x = ws.x[0][0]
        ops.embedding_fwd(ids, self.params["model.embed_tokens.weight"], x)
        t0, t1 = ws.tmpH[0], ws.tmpH[1]
        emb_m = self.shadow.params16["model.embed_tokens.weight"] if bf16 else self.params["model.embed_tokens.weight"]

        # fp32: the q|k|v projection carries per-head RMSNorm + RoPE in its epilogue when its tiles are whole
        fuse_qkv = ((self.fuse_qkv_bf16 if bf16 else self.fuse_qkv) and ops.qkv_fused_ok(ws.layers[0]["h1"], T, QKV)
                    and cfg.head_dim == 64)

        # matmul="split6" / "split9": the attention products run on the bf16 pipe as well (gamer_attn_fwd_split: exact
        # three-way cuts, six piece products); the session variant's key spans and the evaluation re-run path stay on the
        # fp32-MFMA kernels
        split_attn = self.split_attention and self.matmul != "f32" and not bf16

        # (three-product attention) max |v| out of the q|k|v projection's epilogue instead of a pass over the v columns; the cross
        # attention adds its behaviour bias to v afterwards (qknorm_rope_fwd), so only the self attention's v can take it
        h2_now = split_attn and self.h2_attention and self.matmul == "split3" and p_att < 0.75

        # (three-product form) the SwiGLU forward as the epilogue of the experts' gate|up projection (gamer_gemm_desc.sw_hm); A/B switch
        fuse_swiglu_fwd = (not bf16 and self.matmul == "split3" and os.environ.get("GAMER_FUSE_SWIGLU_FWD", "1") != "0")

        def v_amax(qkv_buf):
            return dict(c_amax=(qkv_buf[:, NQ + NKV:], NQ + NKV)) if (h2_now and not bf16 and self.gemm_c_amax) else {}

        def attention(qb, kb, vb, kl_, ql_, empty_, tile_empty_, seed_, ob, lseb, order_, span_):
            if split_attn and (span_ is None or h2_now):
                # (per-query key spans - the session variant - are built for the three-product form; the six-product form sends
                # those calls to the fp32-MFMA kernels below)
                ops.attn_fwd_split(qb, NQ, kb, NKV, vb, QKV, kl_, ql_, empty_, B, S, nq, nkv, scale, p_att, seed_, ob, lseb,
                                   order=order_, h2=h2_now, uniform_len=uniform_len, q_span=span_)
            elif bf16:
                ord16 = (order_[0], order_[2], empty_) if (order_ is not None and span_ is None) else None
                ops.attn_fwd_bf16(qb, NQ, kb, NKV, vb, QKV, kl_, ql_, B, S, nq, nkv, scale, p_att, seed_, ob, lseb,
                                  q_span=span_, order=ord16)
            else:
                ops.attn_fwd(qb, NQ, kb, NKV, vb, QKV, kl_, ql_, empty_, tile_empty_, B, S, nq, nkv, scale, p_att, seed_,
                             ob, lseb, order=order_, uniform_len=uniform_len, q_span=span_)

        for l in range(cfg.num_hidden_layers):
            W, A, xs = self.W[l], ws.layers[l], ws.x[l]
            Wm = self.Wm[l]                          # GEMM operands (fp32 masters or bf16 copies); W: norms, tables
            if hidden_sink is not None:
                hidden_sink.append(xs[0].view(B, S, H).clone())
            # ---- self attention (model.py:204-217) ----
            ops.rmsnorm_fwd(xs[0], W.ln1, eps, A["h1"])
            # (kv_dest: the q|k|v projection and the rotated keys of this layer are written where the caller keeps them - a
            # generation's prompt caches - instead of the workspace's buffers and a copy)
            qkv_s, k_s = kv_dest(l, "self") if kv_dest is not None else (A["qkv"], A["k"])
            if fuse_qkv:
                # per-head RMSNorm + RoPE of q / k in the projection's epilogue (model.py:88-101 in one kernel)
                ops.gemm(A["h1"], H, 1, Wm.self_attn["qkv"], H, 1, qkv_s, QKV, T, QKV, H,
                         qknorm=dict(wq=W.self_attn["qn"], wk=W.self_attn["kn"], eps=eps, cos=cos, sin=sin, q_rot=A["q"],
                                     k_rot=k_s, pos_ids=pos_ids, S=S, nq=nq, nkv=nkv))
            else:
                ops.linear_fwd(A["h1"], H, Wm.self_attn["qkv"], H, qkv_s, QKV, T, QKV, H, **v_amax(qkv_s))
                ops.qknorm_rope_fwd(qkv_s, S, nq, nkv, W.self_attn["qn"], W.self_attn["kn"], eps, cos, sin, A["q"], k_s,
                                    pos_ids=pos_ids)
            if kv_sink is not None:
                kv_sink(l, "self", k_s, qkv_s[:, NQ + NKV:])
            attention(A["q"], k_s, qkv_s[:, NQ + NKV:], r["kl_self"], None, r["empty_self"], r["tile_empty_self"],
                      self._seed(l, 0), A["ao"], A["lse"], None, span_self)
            # o_proj with the residual add + dropout fused into the GEMM epilogue (model.py:149,217)
            ops.gemm(A["ao"], NQ, 1, Wm.self_attn["o"], NQ, 1, xs[1], H, T, H, NQ, resid=xs[0], p_drop=p_res,
                     seed=self._seed(l, 1))
            xcur = xs[1]
            # ---- behaviour-level "cross" attention (model.py:220-235) ----
            if W.cross:
                C, Cm = W.cross_attn, Wm.cross_attn
                ops.rmsnorm_fwd(xs[1], W.ln2, eps, A["h2"])
                qkv_x, k_x = kv_dest(l, "cross") if kv_dest is not None else (A["qkv_c"], A["k_c"])
                # (evaluation: no workspace tensor is declared unchanging - the layers share the buffers - but h2 is read by two GEMMs
                # of this block, the q|k|v projection and the gate: its maximum is kept between them instead of measured twice)
                h2_hold = self._amax.hold(A["h2"]) if (self._amax is not None and not train) else None
                if h2_hold is not None:
                    h2_hold.__enter__()
                if fuse_qkv:
                    ops.gemm(A["h2"], H, 1, Cm["qkv"], H, 1, qkv_x, QKV, T, QKV, H,
                             qknorm=dict(wq=C["qn"], wk=C["kn"], eps=eps, cos=cos, sin=sin, q_rot=A["q_c"], k_rot=k_x,
                                         bias_q=C["bq"], bias_k=C["bk"], bias_v=C["bv"], act_idx=r["act_idx"],
                                         pos_ids=pos_ids, S=S, nq=nq, nkv=nkv))
                else:
                    ops.linear_fwd(A["h2"], H, Cm["qkv"], H, qkv_x, QKV, T, QKV, H)
                    ops.qknorm_rope_fwd(qkv_x, S, nq, nkv, C["qn"], C["kn"], eps, cos, sin, A["q_c"], k_x,
                                        bias_q=C["bq"], bias_k=C["bk"], bias_v=C["bv"], act_idx=r["act_idx"],
                                        pos_ids=pos_ids)
                if kv_sink is not None:
                    kv_sink(l, "cross", k_x, qkv_x[:, NQ + NKV:])
                if (last_row_logits and l == cfg.num_hidden_layers - 1 and span_cross is None and not bf16 and
                        os.environ.get("GAMER_PREFILL_LAST_ROW", "1") != "0"):
                    # prompt pass of a generation, last layer: its keys / values are cached (above), and of its cross block's OUTPUT only
                    # the last position of every sample is still needed - one query row per sample against the prompt keys
                    # (gamer_attn_decode with no generated position), the output projection, gate and residual on B rows
                    rows = torch.arange(B, device=self.device) * S + (S - 1)
                    f32 = dict(dtype=torch.float32, device=self.device)
                    q_l = A["q_c"].index_select(0, rows).contiguous()
                    key_ok = (r["kl_cross"].view(B, S) < r["ql_cross"].view(B, S)[:, S - 1:S]).to(torch.int32).contiguous()
                    uni = (r["empty_cross"].view(B, S)[:, S - 1] != 0).to(torch.int32).contiguous()
                    kg0 = torch.zeros(B, 1, NKV, **f32)
                    ao_l = torch.empty(B, NQ, **f32)
                    ops.attn_decode(q_l, k_x, qkv_x[:, NQ + NKV:], key_ok, kg0, kg0, 0, False, uni, B, 1, S, nq, nkv, scale, ao_l)
                    op_l, gate_l, x_cross_last = torch.empty(B, H, **f32), torch.empty(B, H, **f32), torch.empty(B, H, **f32)
                    ops.linear_fwd(ao_l, NQ, Cm["o"], NQ, op_l, H, B, H, NQ)
                    ops.linear_fwd(A["h2"].index_select(0, rows).contiguous(), H, Cm["gate"], H, gate_l, H, B, H, H)
                    ops.silu_gate_fwd(op_l, gate_l, x_cross_last, resid=xs[1].index_select(0, rows).contiguous())
                    xcur = None                  # (only the last rows of this layer's cross output exist)
                else:
                    x_cross_last = None
                    attention(A["q_c"], k_x, qkv_x[:, NQ + NKV:], r["kl_cross"], r["ql_cross"], r["empty_cross"],
                              r["tile_empty_cross"], self._seed(l, 2), A["ao_c"], A["lse_c"], ws.cross_order, span_cross)
                if x_cross_last is None:
                    ops.linear_fwd(A["ao_c"], NQ, Cm["o"], NQ, A["op_c"], H, T, H, NQ)
                    ops.linear_fwd(A["h2"], H, Cm["gate"], H, A["gate_c"], H, T, H, H)
                if h2_hold is not None:
                    h2_hold.__exit__(None, None, None)
                if x_cross_last is None:
                    # output gate + residual add + dropout in one pass (model.py:147, 235)
                    ops.silu_gate_fwd(A["op_c"], A["gate_c"], xs[2], resid=xs[1], p=p_res, seed=self._seed(l, 3))
                    xcur = xs[2]
            # ---- position-routed SwiGLU experts (model.py:238-241, FFN.py:53-72) ----
            din = W.din
            if last_row_logits and l == cfg.num_hidden_layers - 1:
                # prompt pass of a generation: after the last layer's K/V are cached only the last position of every
                # sample is still needed; its FFN is one dense expert on B rows (the prompt ends with a real token,
                # so the expert is the position's: router.py:83-104)
                rows = torch.arange(B, device=self.device) * S + (S - 1)
                e = (S - 1) % cfg.num_positions + 1
                xl = x_cross_last if (W.cross and x_cross_last is not None) else xcur.index_select(0, rows).contiguous()
                f32 = dict(dtype=torch.float32, device=self.device)
                hin = torch.empty(B, din, **f32)
                ops.rmsnorm_fwd(xl, W.ln3, eps, hin, din)
                if W.inject:
                    ops.rowtable_fwd(W.beh, r["beh_idx"].view(-1).index_select(0, rows).contiguous(), hin, din, H)
                gu, hm = torch.empty(B, 2 * I, **f32), torch.empty(B, I, **f32)
                ops.linear_fwd(hin, din, W.gu[e * 2 * I:(e + 1) * 2 * I], din, gu, 2 * I, B, 2 * I, din)
                ops.swiglu_fwd_ld(gu, 2 * I, B, I, 0.0, 0, hm)
                x_last = torch.empty(B, H, **f32)
                ops.gemm(hm, I, 1, W.down[e * H:(e + 1) * H], I, 1, x_last, H, B, H, I, resid=xl)
                break
            grp = dict(groups=E, group_offsets=ws.offsets)
            if W.inject and self.split_inject:
                # K = 256 of the 320 input columns; the embedding columns' share comes from the (expert, behaviour) table
                hin = A["hin"].view(-1)[:T * H].view(T, H)
                ops.rmsnorm_fwd(xcur, W.ln3, eps, hin, H, ws.slot)
                tb = ws._buf((f"l{l}_" if train else "l_") + "inject_tbl", (E * NB1, 2 * I))
                ops.inject_table_fwd(W.beh, W.gu, din, H, E, 2 * I, tb)
                if fuse_swiglu_fwd:
                    # one call: gate|up AND hm (the SwiGLU forward as the projection's epilogue; rows grouped by (expert, behaviour))
                    ops.gemm(hin, H, 1, Wm.gu, din, 1, A["gu"], 2 * I, T, 2 * I, H, strideB=2 * I * din, groups=E * NB1,
                             group_offsets=ws.grp_offsets, group_div=NB1, p_drop=p_res, seed=self._seed(l, 4),
                             swiglu_fwd=(A["hm"], tb, ws.row_group))
                else:
                    ops.linear_fwd(hin, H, Wm.gu, din, A["gu"], 2 * I, T, 2 * I, H, strideB=2 * I * din, **grp)
                    ops.swiglu_fwd_ld_tbl(A["gu"], 2 * I, T, I, p_res, self._seed(l, 4), A["hm"], tb, ws.row_group)
            else:
                ops.rmsnorm_fwd(xcur, W.ln3, eps, A["hin"], din, ws.slot)
                if W.inject:
                    ops.rowtable_fwd(W.beh, r["beh_idx"], A["hin"], din, H, ws.slot)
                # gate_proj and up_proj of the position's expert in ONE grouped GEMM against the stacked [2 I, din] weight (FFN.py:25-27:
                # both read the same input): A["gu"][:, :I] = gate, [:, I:] = up
                if fuse_swiglu_fwd and din == H:
                    ops.gemm(A["hin"], din, 1, Wm.gu, din, 1, A["gu"], 2 * I, T, 2 * I, din, strideB=2 * I * din, p_drop=p_res,
                             seed=self._seed(l, 4), swiglu_fwd=(A["hm"], None, None), **grp)
                else:
                    ops.linear_fwd(A["hin"], din, Wm.gu, din, A["gu"], 2 * I, T, 2 * I, din, strideB=2 * I * din, **grp)
                    ops.swiglu_fwd_ld(A["gu"], 2 * I, T, I, p_res, self._seed(l, 4), A["hm"])
            xnext = ws.x[l + 1][0] if l + 1 < cfg.num_hidden_layers else ws.x_final
            # down projection: rows are in expert-sorted order, the epilogue scatters them back to token
            # order through perm while adding the residual and applying dropout (FFN.py:25-27, model.py:241)
            ops.gemm(A["hm"], I, 1, Wm.down, I, 1, xnext, H, T, H, I, strideB=H * I, resid=xcur, row_map=ws.perm,
                     p_drop=p_res, seed=self._seed(l, 5), **grp)
        # ---- final norm, tied head, temperature CE (model.py:869,1001,904-922) ----
        V = cfg.vocab_size
        if last_row_logits:
            xn = torch.empty_like(x_last)
            ops.rmsnorm_fwd(x_last, self.params["model.norm.weight"], eps, xn)
            small = torch.empty(B, ws.ldl, dtype=torch.float32, device=self.device)
            ops.linear_fwd(xn, H, self.params["model.embed_tokens.weight"], H, small, ws.ldl, B, V, H)
            self._saved = None
            self.last_logits_buf = small               # [B, ldl] (columns >= V are padding), what DecodeSession scores
            return None, small.view(B, 1, ws.ldl)[:, :, :V]
        ops.rmsnorm_fwd(ws.x_final, self.params["model.norm.weight"], eps, ws.xn)
        if hidden_sink is not None:
            hidden_sink.append(ws.xn.view(B, S, H).clone())
        # with labels the logits leave the head GEMM already divided by the temperature (its alpha; model.py:913 divides them in place):
        # the loss kernel then only reads them (fp32 engine; the bf16 head rounds to bf16 first, as the reference's autocast does)
        head_alpha = (1.0 / self.temperature) if (lab is not None and ws.logits.dtype == torch.float32) else 1.0
        ops.linear_fwd(ws.xn, H, emb_m, H, ws.logits, ws.ldl, T, V, H, alpha=head_alpha)
        loss = None
        if lab is not None:
            ops.ce_fwd(ws.logits, ws.ldl, lab, V, 1.0 if head_alpha != 1.0 else self.temperature, IGNORE_INDEX, ws.lse_ce,
                       ws.row_loss, ws.loss_sum, ws.count)
            if torch.is_tensor(num_items_in_batch):
                # a device scalar (the label count of the window summed over ranks): never read on the host
                num_items_in_batch = num_items_in_batch.to(self.device, torch.float32).reshape(1)
                loss = ws.loss_sum[0] / num_items_in_batch[0]
            elif num_items_in_batch is not None:
                loss = ws.loss_sum[0] / float(num_items_in_batch)
            else:
                loss = ws.loss_sum[0] / ws.count[0]
        self._saved = dict(ids=ids, labels=lab, num_items=num_items_in_batch, train=train, p_res=p_res, p_att=p_att,
                           B=B, S=S, dropout_step=self.dropout_step)
        logits = ws.logits.view(B, S, ws.ldl)[:, :, :V]
        return loss, logits

    def check_inputs(self):
        """Host-synchronising validation (what the reference would raise on): unknown behaviour tokens."""
        n = int(self.ws.router["bad_token"].item())
        if n:
            raise IndexError(f"{n} item(s) start with a token that is not in config.behavior_maps "
                             "(the reference fails with an embedding IndexError, router.py:170-171)")
        n = int(self.ws.bad_label.item())
        if n:
            raise IndexError(f"{n} label(s) outside [0, vocab_size={self.cfg.vocab_size}) that are not -100 "
                             "(nn.CrossEntropyLoss raises 'Target out of bounds' in the reference)")
        if self.variant == "session" and self.ws.session is not None:
            n = int(self.ws.session["violations"].item())
            if n:
                raise ValueError(f"{n} row(s) with session ids that decrease along the sequence or RoPE positions "
                                 "outside [0, S): the session masks are built as causal key spans "
                                 "(gamer_session_spans), which needs the dataset's layout (SMB_dataset.py:194-222)")

    # ------------------------------------------------------------------------------------------
    def zero_grad(self):
        ops.fill(self.flat_g, 0.0)

    def backward(self, dloss: float = 1.0, layer_done=None, dloss_dev: Optional[torch.Tensor] = None):
        with ops.deterministic(True if self.deterministic else ops.DETERMINISTIC_WGRAD, bf16=True if self.deterministic else None):
            return self._backward(dloss, layer_done, dloss_dev)

    @ops.scoped_f32_matmul(lambda self, *a: self.matmul, lambda self, *a: self._planes())
    @ops.scoped_amax(lambda self, *a: self._amax)
    def _backward(self, dloss: float = 1.0, layer_done=None, dloss_dev: Optional[torch.Tensor] = None):
        """Accumulates d(loss)*dloss into the flat gradient buffer (call zero_grad() first for a fresh
        window).  Needs a forward(..., labels=..., train=True) before it.  ``dloss_dev``: fp32 device scalar that
        multiplies ``dloss`` (the module path hands autograd's incoming gradient over without reading it on the host).  ``layer_done(l)`` is called as
        soon as every kernel writing layer l's weight gradients has been enqueued (data-parallel
        all-reduce overlap, gamer_amd.dp)."""
        sv = self._saved
        if sv is None or not sv["train"] or sv["labels"] is None:
            raise RuntimeError("backward() needs forward(train=True, labels=...) first")
        cfg, ws = self.cfg, self.ws
        B, S = sv["B"], sv["S"]
        span_self = span_cross = pos_ids = None
        if self.variant == "session":
            span_self, span_cross, pos_ids = ws.session["span_self"], ws.session["span_cross"], ws.session["pos_ids"]
        T, H = B * S, cfg.hidden_size
        nq, nkv, dh, I, E = (cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size,
                             cfg.num_experts)
        NQ, NKV = nq * dh, nkv * dh
        QKV = NQ + 2 * NKV
        NB1 = cfg.num_behavior + 1
        eps = float(cfg.rms_norm_eps)
        p_res, p_att = sv["p_res"], sv["p_att"]
        saved_step = self.dropout_step
        self.dropout_step = sv["dropout_step"]          # regenerate exactly the forward's masks
        r = ws.router
        cos, sin = self.rope(S)
        scale = float(dh) ** -0.5
        V = cfg.vocab_size
        emb = self.params["model.embed_tokens.weight"]
        demb = self.grads["model.embed_tokens.weight"]
        NP = ws.norm_partial[-1]                    # scratch table (rowtable_bwd); the norm backward takes the others in turn
        norm_dws: List[torch.Tensor] = []
        # delta = dO . O of the attention backward comes out of the o_proj dgrad GEMM (row-dot epilogue) when the dS-spill
        # path is in use and every tile of that GEMM is full; otherwise gamer_attn_bwd computes it itself
        bf16 = self.dtype == "bf16"
        split_attn = self.split_attention and self.matmul != "f32" and not bf16
        fuse_delta = (ws.ds_work is not None or bf16 or split_attn) and T % 128 == 0 and NQ % 128 == 0
        # (three-product attention) max |dO| out of the o_proj input-gradient GEMM's epilogue instead of a pass over dO
        do_amax = (dict(c_amax=(ws.dao, 0)) if (split_attn and self.h2_attention and self.matmul == "split3" and p_att < 0.75 and
                                                   self.gemm_c_amax) else {})

        fuse_swiglu_bwd = (bf16 or self.matmul == "split3") and self.fuse_swiglu_bwd

        def dgrad(dy, lddy, Wf, Wt, ldw, dx, lddx, n_out, k_in, **kw):
            """dx[T,k_in] (+)= dy[T,n_out] @ W[n_out,k_in]: fp32 reads W itself (row-contiguous B operand), bf16 the
            transposed copy Wt [k_in, n_out] (k-contiguous on both sides); strideB is the same element count."""
            if bf16:
                ops.linear_dgrad_t(dy, lddy, Wt, Wt.shape[1], dx, lddx, T, n_out if Wt.shape[1] == n_out else Wt.shape[1],
                                   k_in, **kw)
            else:
                ops.linear_dgrad(dy, lddy, Wf, ldw, dx, lddx, T, n_out, k_in, **kw)

        def attention_bwd(qb, kb, vb, ob, lseb, kl_, ql_, empty_, tile_empty_, seed_, order_, span_):
            h2_bwd = self.h2_attention and self.matmul == "split3" and p_att < 0.75
            if split_attn and (span_ is None or h2_bwd):
                ops.attn_bwd_split(qb, NQ, kb, NKV, vb, QKV, ob, ws.dao, lseb, kl_, ql_, empty_, tile_empty_, B, S, nq, nkv, scale,
                                   p_att, seed_, ws.delta, ws.dq, NQ, ws.dk, NKV, ws.dqkv[:, NQ + NKV:], QKV, order=order_,
                                   delta_ready=fuse_delta, dv_of=ws.dqkv,   # recompute form: measured faster than its dS spill
                                   h2=h2_bwd, q_span=span_)
            elif bf16:
                ord16 = (order_[0], order_[2], empty_) if (order_ is not None and span_ is None) else None
                ops.attn_bwd_bf16(qb, NQ, kb, NKV, vb, QKV, ob, ws.dao, lseb, kl_, ql_, B, S, nq, nkv, scale, p_att, seed_,
                                  ws.delta, ws.dq, NQ, ws.dk, NKV, ws.dqkv[:, NQ + NKV:], QKV, q_span=span_,
                                  delta_ready=fuse_delta, order=ord16)
            else:
                ops.attn_bwd(qb, NQ, kb, NKV, vb, QKV, ob, ws.dao, lseb, kl_, ql_, empty_, tile_empty_, B, S, nq, nkv, scale,
                             p_att, seed_, ws.delta, ws.dq, NQ, ws.dk, NKV, ws.dqkv[:, NQ + NKV:], QKV, order=order_,
                             ds_work=ws.ds_work, q_span=span_, delta_ready=fuse_delta and ws.ds_work is not None)

        def norm_bwd(xin, w, dy, lddy, dw, accumulate_dx, dy_rows=None, branch=None):
            """``branch`` = (seed, rows): the residual branch that consumes the updated dx next; its input gradient
            dropout_mask(seed) * dx goes to t0 in the same pass (t0 must not be ``dy``)."""
            part = ws.norm_partial[len(norm_dws)]       # dw itself is summed at the end of the pass (one launch for all norms:
            norm_dws.append(dw)                         # they sit in the tail bucket of the gradient all-reduce)
            if branch is None:
                ops.rmsnorm_bwd(xin, w, dy, lddy, eps, ws.dx, part, accumulate_dx, dy_rows)
            else:
                ops.rmsnorm_bwd(xin, w, dy, lddy, eps, ws.dx, part, accumulate_dx, dy_rows, mask_out=ws.tmpH[0],
                                mask_rows=branch[1], p=p_res, seed=branch[0])

        # ---- loss -> logits -> final norm ----
        if torch.is_tensor(sv["num_items"]):
            ops.ce_bwd(ws.logits, ws.ldl, sv["labels"], V, self.temperature, IGNORE_INDEX, ws.lse_ce, sv["num_items"], 0.0,
                       dloss, dloss_dev)
        elif sv["num_items"] is not None:
            ops.ce_bwd(ws.logits, ws.ldl, sv["labels"], V, self.temperature, IGNORE_INDEX, ws.lse_ce, None,
                       float(sv["num_items"]), dloss, dloss_dev)
        else:
            ops.ce_bwd(ws.logits, ws.ldl, sv["labels"], V, self.temperature, IGNORE_INDEX, ws.lse_ce, ws.count, 0.0,
                       dloss, dloss_dev)
        import contextlib
        # (matmul="split3") a gradient read by the weight- and the input-gradient GEMM of a site is measured once
        hold = self._amax.hold if self._amax is not None else (lambda *t: contextlib.nullcontext())
        t0, t1, t2, t3 = ws.tmpH
        L = cfg.num_hidden_layers
        with hold(ws.logits):
            ops.linear_wgrad(ws.logits, ws.ldl, ws.xn, H, demb, H, T, V, H)
            dgrad(ws.logits, ws.ldl, emb, self.shadow.t("model.embed_tokens.weight") if bf16 else None, H, t3, H, V, H)
        # every norm backward that completes dx also emits t0 = mask * dx for the branch that reads dx next
        norm_bwd(ws.x_final, self.params["model.norm.weight"], t3, H, self.grads["model.norm.weight"], False,
                 branch=(self._seed(L - 1, 5), ws.slot))

        for l in reversed(range(cfg.num_hidden_layers)):
            W, G, A, xs = self.W[l], self.G[l], ws.layers[l], ws.x[l]
            WT = self.WT[l] if bf16 else None
            din = W.din
            grp = dict(groups=E, group_offsets=ws.offsets)
            xlast = xs[2] if W.cross else xs[1]
            # ---- experts ----   (t0 = d out_sorted, written by the norm backward that completed dx)
            split_inj = W.inject and self.split_inject
            tb = ws._buf(f"l{l}_inject_tbl", (E * NB1, 2 * I)) if split_inj else None
            with hold(t0):
                ops.linear_wgrad(t0, H, A["hm"], I, G.down, I, T, H, I, strideC=H * I, **grp)
                if fuse_swiglu_bwd and split_inj:
                    # (rows grouped by (expert, behaviour): a tile's rows share one table row; NB1 consecutive groups share an expert)
                    ops.gemm(t0, H, 1, W.down, 1, I, ws.dhm, I, T, I, H, strideB=H * I, p_drop=p_res, seed=self._seed(l, 4),
                             swiglu_bwd=(A["gu"], 2 * I), groups=E * NB1, group_offsets=ws.grp_offsets, group_div=NB1, sw_tbl=tb)
                elif fuse_swiglu_bwd:
                    # the down projection's input gradient with the SwiGLU backward in its epilogue: d(hm) = t0 W_down never goes
                    # to memory, A["gu"] <- d gate | d up (gamer_gemm_desc.sw_gu; ws.dhm is only the descriptor's C)
                    if bf16:
                        ops.linear_dgrad_t(t0, H, WT.down, WT.down.shape[1], ws.dhm, I, T, H, I, strideB=H * I, p_drop=p_res,
                                           seed=self._seed(l, 4), swiglu_bwd=(A["gu"], 2 * I), **grp)
                    else:
                        ops.gemm(t0, H, 1, W.down, 1, I, ws.dhm, I, T, I, H, strideB=H * I, p_drop=p_res, seed=self._seed(l, 4),
                                 swiglu_bwd=(A["gu"], 2 * I), **grp)
                else:
                    dgrad(t0, H, W.down, WT.down if bf16 else None, I, ws.dhm, I, H, I, strideB=H * I, **grp)
            if not fuse_swiglu_bwd:
                if split_inj:
                    ops.swiglu_bwd_ld_tbl(A["gu"], 2 * I, T, I, ws.dhm, p_res, self._seed(l, 4), tb, ws.row_group)
                else:
                    ops.swiglu_bwd_ld(A["gu"], 2 * I, T, I, ws.dhm, p_res, self._seed(l, 4))    # gu <- d gate | d up
            if split_inj:
                hin = A["hin"].view(-1)[:T * H].view(T, H)
                dhin = ws.dhin.view(-1)[:T * H].view(T, H)
                with hold(A["gu"]):
                    ops.linear_wgrad(A["gu"], 2 * I, hin, H, G.gu, din, T, 2 * I, H, strideC=2 * I * din, **grp)
                    dgrad(A["gu"], 2 * I, W.gu, None, din, dhin, H, 2 * I, H, strideB=2 * I * din, **grp)
                # the embedding columns' weight gradient and the embedding gradient from the (expert, behaviour) sums of d(gate|up)
                seg = ws._buf("inject_segsum", (E * NB1, 2 * I))
                seg_ws = ws._buf("inject_seg_ws", (ops.segment_colsum_ws_floats(T, 2 * I, E * NB1),))
                ops.segment_colsum(A["gu"], 2 * I, T, 2 * I, ws.grp_offsets, E * NB1, seg_ws, seg)
                ops.inject_table_bwd(seg, W.beh, W.gu, din, H, E, 2 * I, G.gu, G.beh,
                                     ws._buf("inject_scratch", (NB1 * E * cfg.behavior_embedding_dim,)))
                norm_bwd(xlast, W.ln3, dhin, H, G.ln3, True, ws.slot, branch=None if W.cross else (self._seed(l, 1), None))
            else:
                with hold(A["gu"]):
                    ops.linear_wgrad(A["gu"], 2 * I, A["hin"], din, G.gu, din, T, 2 * I, din, strideC=2 * I * din, **grp)
                    dgrad(A["gu"], 2 * I, W.gu, WT.gu if bf16 else None, din, ws.dhin, din, 2 * I, din, strideB=2 * I * din, **grp)
                if W.inject:
                    ops.rowtable_bwd(ws.dhin, din, H, r["beh_idx"], G.beh, ws.slot, partial=NP)
                norm_bwd(xlast, W.ln3, ws.dhin, din, G.ln3, True, ws.slot,
                         branch=None if W.cross else (self._seed(l, 1), None))
            # ---- cross attention ----
            if W.cross:
                C, GC = W.cross_attn, G.cross_attn
                CT = WT.cross_attn if bf16 else dict(o=None, gate=None, qkv=None)
                # dropout mask of the residual add + gate backward in one pass: t1 = d op, t2 = d gate
                ops.silu_gate_bwd(A["op_c"], A["gate_c"], ws.dx, t1, t2, p=p_res, seed=self._seed(l, 3))
                hold_t2 = hold(t2)                       # (t2 is read again by its input-gradient GEMM below)
                hold_t2.__enter__()
                with hold(t1):
                    ops.linear_wgrad(t1, H, A["ao_c"], NQ, GC["o"], NQ, T, H, NQ)
                    ops.linear_wgrad(t2, H, A["h2"], H, GC["gate"], H, T, H, H)
                    dgrad(t1, H, C["o"], CT["o"], NQ, ws.dao, NQ, H, NQ,
                          rowdot=(A["ao_c"], ws.delta, S) if fuse_delta else None, **do_amax)
                attention_bwd(A["q_c"], A["k_c"], A["qkv_c"][:, NQ + NKV:], A["ao_c"], A["lse_c"], r["kl_cross"],
                              r["ql_cross"], r["empty_cross"], r["tile_empty_cross"], self._seed(l, 2), ws.cross_order,
                              span_cross)
                ops.qknorm_rope_bwd(A["qkv_c"], ws.dq, ws.dk, S, nq, nkv, C["qn"], C["kn"], eps, cos, sin, ws.dqkv,
                                    GC["qn"], GC["kn"], bias_q=C["bq"], bias_k=C["bk"], act_idx=r["act_idx"], nb1=NB1,
                                    dbias_q=GC["bq"], dbias_k=GC["bk"], dbias_v=GC["bv"], pos_ids=pos_ids,
                                    partial=ws.qk_partial)
                with hold(ws.dqkv):
                    ops.linear_wgrad(ws.dqkv, QKV, A["h2"], H, GC["qkv"], H, T, QKV, H)
                    dgrad(ws.dqkv, QKV, C["qkv"], CT["qkv"], H, t3, H, QKV, H)
                dgrad(t2, H, C["gate"], CT["gate"], H, t3, H, H, H, accumulate=True)
                hold_t2.__exit__(None, None, None)
                norm_bwd(xs[1], W.ln2, t3, H, G.ln2, True, branch=(self._seed(l, 1), None))
            # ---- self attention ----
            SA, GS = W.self_attn, G.self_attn
            ST = WT.self_attn if bf16 else dict(o=None, qkv=None)
            # (t0 = mask * dx of this branch, written by the norm backward above)
            with hold(t0):
                ops.linear_wgrad(t0, H, A["ao"], NQ, GS["o"], NQ, T, H, NQ)
                dgrad(t0, H, SA["o"], ST["o"], NQ, ws.dao, NQ, H, NQ, rowdot=(A["ao"], ws.delta, S) if fuse_delta else None,
                      **do_amax)
            attention_bwd(A["q"], A["k"], A["qkv"][:, NQ + NKV:], A["ao"], A["lse"], r["kl_self"], None, r["empty_self"],
                          r["tile_empty_self"], self._seed(l, 0), None, span_self)
            ops.qknorm_rope_bwd(A["qkv"], ws.dq, ws.dk, S, nq, nkv, SA["qn"], SA["kn"], eps, cos, sin, ws.dqkv, GS["qn"],
                                GS["kn"], pos_ids=pos_ids, partial=ws.qk_partial)
            with hold(ws.dqkv):
                ops.linear_wgrad(ws.dqkv, QKV, A["h1"], H, GS["qkv"], H, T, QKV, H)
                dgrad(ws.dqkv, QKV, SA["qkv"], ST["qkv"], H, t3, H, QKV, H)
            norm_bwd(xs[0], W.ln1, t3, H, G.ln1, True, branch=(self._seed(l - 1, 5), ws.slot) if l > 0 else None)
            if layer_done is not None:
                layer_done(l)
        self._check_deterministic_embedding()        # (raised in __init__ already: nothing has been written when it fires there)
        if self.ordered_embedding_grad and V <= 8191:
            # the scatter-add of the embedding gradient in a fixed order: a stable counting sort of the tokens by id, then sums in
            # token order (no float atomics: with the ordered weight gradients and table gradients every gradient of the step
            # has the same bits on every run); pad and out-of-range ids are skipped (padding_idx, model.py:263)
            ops.embedding_bwd_ordered(sv["ids"], ws.dx, cfg.pad_token_id, demb)
        else:
            ops.embedding_bwd(sv["ids"], ws.dx, cfg.pad_token_id, demb)
        key = tuple(d.data_ptr() for d in norm_dws)
        if self._norm_out_table is None or self._norm_out_table[0] != key:
            self._norm_out_table = (key, torch.tensor(key, dtype=torch.int64, device=self.device))
        ops.colsum_reduce_batched(ws.norm_partial, len(norm_dws), self._norm_out_table[1], accumulate=True)
        self.dropout_step = saved_step

    # ------------------------------------------------------------------------------------------
    def optimizer_step(self, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, max_norm=1.0,
                       grad_scale: float = 1.0):
        """clip_grad_norm_(max_norm) + AdamW on the flat buffers (HF Trainer defaults,
        ref:SeqRec/tasks/train_SMB_decoder.py:396-428).  ``grad_scale`` multiplies the gradient first
        (1/world_size after a sum all-reduce, 1/accumulation steps)."""
        if self.flat_m is None:
            self.flat_m = torch.zeros_like(self.flat_p)
            self.flat_v = torch.zeros_like(self.flat_p)
        self.opt_step += 1
        ops.sumsq(self.flat_g, self.sumsq_partial)
        ops.adamw(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.layout.n_decay, float(lr), beta1, beta2, eps,
                  weight_decay, self.opt_step, float(max_norm), float(grad_scale), self.sumsq_partial, self.grad_norm)

    def train_step(self, batch: Dict[str, torch.Tensor], lr: float, num_items_in_batch=None, reducer=None, **opt):
        """forward + backward (+ data-parallel gradient sum) + update on one micro-batch.

        With a ``gamer_amd.dp.GradAllReducer`` the label count is summed over ranks before the
        backward (loss = sum CE / global count, HF average_tokens_across_devices) and the per-layer
        gradient buckets are all-reduced while the remaining layers' backward runs."""
        loss, _ = self.forward(batch["input_ids"], batch.get("attention_mask"), batch.get("actions"),
                               labels=batch["labels"], num_items_in_batch=num_items_in_batch, train=True,
                               session_ids=batch.get("session_ids"),
                               extended_session_ids=batch.get("extended_session_ids"))
        self.zero_grad()
        if reducer is not None and num_items_in_batch is None:
            from .dp import all_reduce_scalar_
            all_reduce_scalar_(self.ws.count)
        self.backward(1.0, layer_done=reducer.layer_done if reducer is not None else None)
        if reducer is not None:
            reducer.finish()
        self.optimizer_step(lr, **opt)
        return loss

    def train_window(self, micro_batches, lr: float, n_items=None, reducer=None, **opt):
        """One optimizer step over a gradient-accumulation window (HF Trainer with gradient_accumulation_steps =
        len(micro_batches), ref:SeqRec/tasks/train_SMB_decoder.py:405, README recipe 4): the loss of every micro-batch is
        sum CE / ``n_items`` with ``n_items`` the label count of the WHOLE window (HF ``num_items_in_batch``; counted
        here when None), summed over ranks on the device when a ``GradAllReducer`` is given
        (``average_tokens_across_devices``); gradients accumulate in the flat buffer and the per-layer buckets are
        all-reduced during the LAST micro-batch's backward only.  Returns the list of micro-batch losses (device)."""
        if n_items is None:
            n_items = sum(int((b["labels"][:, 1:] != IGNORE_INDEX).sum()) for b in micro_batches)
        if not torch.is_tensor(n_items):
            n_items = torch.tensor([float(n_items)], dtype=torch.float32, device=self.device)
        else:
            n_items = n_items.to(self.device, torch.float32).reshape(1)
        if reducer is not None:
            from .dp import all_reduce_scalar_
            all_reduce_scalar_(n_items)
        self.zero_grad()
        losses = []
        for a, b in enumerate(micro_batches):
            loss, _ = self.forward(b["input_ids"], b.get("attention_mask"), b.get("actions"), labels=b["labels"],
                                   num_items_in_batch=n_items, train=True, session_ids=b.get("session_ids"),
                                   extended_session_ids=b.get("extended_session_ids"))
            last = a == len(micro_batches) - 1
            self.backward(1.0, layer_done=reducer.layer_done if (reducer is not None and last) else None)
            losses.append(loss)
        if reducer is not None:
            reducer.finish()
        self.optimizer_step(lr, **opt)
        return losses
