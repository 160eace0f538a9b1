"""``SeqRec.modules.layers.transformer`` on the HIP path.

Same nn.Module surface and parameter names as the reference's post-LN BERT-style encoder used by its
discriminative baselines (ref:SeqRec/modules/layers/transformer.py:12-183): ``MultiHeadAttention`` (query / key /
value / dense Linear layers with bias, additive attention mask, dropout on the probabilities and on the output,
LayerNorm(h + x)), ``FeedForward`` (dense_1, activation, dense_2), ``TransformerEncoderLayer``,
``TransformerEncoder``.  The forward and the backward of one encoder layer run as HIP kernels through the C ABI
(GEMMs: gamer_gemm_f32; bias/activation, LayerNorm, dense attention: csrc/modules.hip) behind one
``torch.autograd.Function``; there is no PyTorch fallback.

Reference behaviour kept on purpose: ``FeedForward`` creates its LayerNorm / dropout when ``residual`` is True but
applies them only ``if not self.residual`` (transformer.py:96-98 vs :116-118), so the layer every model builds
(residual=True) returns ``dense_2(act(dense_1(x)))`` with no residual and no norm - its ``LayerNorm`` parameters
exist in the state dict, receive no gradient - and residual=False cannot run upstream at all (AttributeError).
"""
from __future__ import annotations

import copy
import math
from typing import Callable, Optional

import torch
from torch import nn
from torch.nn import functional as F

from . import ops

_N_PARTIAL = 256
_ACT_BY_FN = {F.relu: "relu", F.gelu: "gelu", F.silu: "swish", F.tanh: "tanh", torch.tanh: "tanh",
              F.sigmoid: "sigmoid", torch.sigmoid: "sigmoid", F.elu: "elu"}


def _act_code(activation) -> int:
    if isinstance(activation, str):
        if activation not in ops.ACTIVATIONS:
            raise KeyError(activation)
        return ops.ACTIVATIONS[activation]
    if activation in _ACT_BY_FN:
        return ops.ACTIVATIONS[_ACT_BY_FN[activation]]
    raise NotImplementedError(f"activation {activation!r}: only relu/gelu/swish/tanh/sigmoid/elu run on the HIP path")


class _SeedCounter:
    value = 0x51ED

    @classmethod
    def next(cls) -> int:
        cls.value += 1
        return cls.value


class _EncoderLayerFn(torch.autograd.Function):
    """One TransformerEncoderLayer: forward keeps the activations the hand-written backward needs."""

    @staticmethod
    @ops.scoped_f32_matmul(lambda *a: "f32")
    def forward(ctx, x, mask, meta, wq, bq, wk, bk, wv, bv, wd, bd, ln1w, ln1b, w1, b1, w2, b2):
        if not x.is_cuda:
            raise RuntimeError("gamer_amd.modules runs on the HIP device only (no CPU fallback)")
        B, S, D = x.shape
        T = B * S
        H, dff, act, p, eps, training = (meta["heads"], meta["dff"], meta["act"], meta["dropout"], meta["eps"],
                                         meta["training"])
        p = p if training else 0.0
        dh = D // H
        f32 = dict(dtype=torch.float32, device=x.device)
        xf = x.reshape(T, D).contiguous().float()
        seeds = [_SeedCounter.next() for _ in range(2)]
        # q, k, v projections as one GEMM on the concatenated weights
        wqkv = torch.cat([wq, wk, wv], 0).contiguous()
        bqkv = torch.cat([bq, bk, bv], 0).contiguous()
        qkv = torch.empty(T, 3 * D, **f32)
        ops.linear_fwd(xf, D, wqkv, D, qkv, 3 * D, T, 3 * D, D)
        ops.bias_act_fwd(qkv, bqkv, 0)
        ctxv = torch.empty(T, D, **f32)
        lse = torch.empty(B, H, S, **f32)
        scale = math.sqrt(1.0 / float(dh))
        m = mask.float().contiguous() if mask is not None else None
        ops.attn_dense_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], m, B, S, H, dh, scale, p, seeds[0], ctxv, lse)
        h = torch.empty(T, D, **f32)
        ops.linear_fwd(ctxv, D, wd, D, h, D, T, D, D)
        ops.bias_act_fwd(h, bd, 0)
        v1 = torch.empty(T, D, **f32)
        ops.residual_dropout_fwd(xf, h, p, seeds[1], None, v1)              # x + dropout(dense(context))
        y1 = torch.empty(T, D, **f32)
        mean1, rstd1 = torch.empty(T, **f32), torch.empty(T, **f32)
        ops.layernorm_fwd(v1, None, ln1w, ln1b, eps, None, y1, mean1, rstd1)
        # feed forward
        pre1 = torch.empty(T, dff, **f32)
        a1 = torch.empty(T, dff, **f32)
        ops.linear_fwd(y1, D, w1, D, pre1, dff, T, dff, D)
        ops.bias_act_fwd(pre1, b1, act, a1)
        f2 = torch.empty(T, D, **f32)
        ops.linear_fwd(a1, dff, w2, dff, f2, D, T, D, dff)
        ops.bias_act_fwd(f2, b2, 0)
        ctx.meta = dict(meta, p=p, seeds=seeds, scale=scale, shape=(B, S, D))
        ctx.mask = m
        ctx.save_for_backward(xf, wqkv, qkv, ctxv, lse, wd, v1, mean1, rstd1, ln1w, y1, w1, pre1, a1, w2)
        return f2.view(B, S, D)

    @staticmethod
    @ops.scoped_f32_matmul(lambda *a: "f32")
    def backward(ctx, dout):
        xf, wqkv, qkv, ctxv, lse, wd, v1, mean1, rstd1, ln1w, y1, w1, pre1, a1, w2 = ctx.saved_tensors
        mt = ctx.meta
        B, S, D = mt["shape"]
        T, H, dff, act, p, seeds = B * S, mt["heads"], mt["dff"], mt["act"], mt["p"], mt["seeds"]
        dh = D // H
        f32 = dict(dtype=torch.float32, device=xf.device)
        part = torch.empty(_N_PARTIAL, max(3 * D, dff), **f32)
        part2 = torch.empty(_N_PARTIAL, D, **f32)

        def colsum(partial_view, n):
            out = torch.empty(n, **f32)
            ops.colsum_reduce(partial_view, out)
            return out
        g = dout.reshape(T, D).contiguous().float().clone()
        # dense_2
        pb = part[:, :D].contiguous()
        ops.bias_act_bwd(None, g, 0, g, pb)
        db2 = colsum(pb, D)
        dw2 = torch.zeros_like(w2)
        ops.linear_wgrad(g, D, a1, dff, dw2, dff, T, D, dff)
        da1 = torch.empty(T, dff, **f32)
        ops.linear_dgrad(g, D, w2, dff, da1, dff, T, D, dff)
        # activation + dense_1
        pb = part[:, :dff].contiguous()
        ops.bias_act_bwd(pre1, da1, act, da1, pb)
        db1 = colsum(pb, dff)
        dw1 = torch.zeros_like(w1)
        ops.linear_wgrad(da1, dff, y1, D, dw1, D, T, dff, D)
        dy1 = torch.empty(T, D, **f32)
        ops.linear_dgrad(da1, dff, w1, D, dy1, D, T, dff, D)
        # LayerNorm(x + dropout(h))
        dv1 = torch.empty(T, D, **f32)
        pw, pb = part[:, :D].contiguous(), part2
        ops.layernorm_bwd(v1, ln1w, mean1, rstd1, dy1, dv1, pw, pb)
        dln1w, dln1b = colsum(pw, D), colsum(pb, D)
        dh_ = torch.empty(T, D, **f32)
        ops.residual_dropout_bwd(dv1, p, seeds[1], dh_)                        # dv1 stays = d x (residual branch)
        pb = part[:, :D].contiguous()
        ops.bias_act_bwd(None, dh_, 0, dh_, pb)
        dbd = colsum(pb, D)
        dwd = torch.zeros_like(wd)
        ops.linear_wgrad(dh_, D, ctxv, D, dwd, D, T, D, D)
        dctx = torch.empty(T, D, **f32)
        ops.linear_dgrad(dh_, D, wd, D, dctx, D, T, D, D)
        dqkv = torch.empty(T, 3 * D, **f32)
        ops.attn_dense_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ctx.mask, B, S, H, dh, mt["scale"], p, seeds[0],
                           ctxv, dctx, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:])
        pb = part[:, :3 * D].contiguous()
        ops.bias_act_bwd(None, dqkv, 0, dqkv, pb)
        dbqkv = colsum(pb, 3 * D)
        dwqkv = torch.zeros_like(wqkv)
        ops.linear_wgrad(dqkv, 3 * D, xf, D, dwqkv, D, T, 3 * D, D)
        ops.linear_dgrad(dqkv, 3 * D, wqkv, D, dv1, D, T, 3 * D, D, accumulate=True)     # dx = dv1 + dqkv Wqkv
        dx = dv1.view(B, S, D)
        return (dx, None, None, dwqkv[:D], dbqkv[:D], dwqkv[D:2 * D], dbqkv[D:2 * D], dwqkv[2 * D:], dbqkv[2 * D:], dwd,
                dbd, dln1w, dln1b, dw1, db1, dw2, db2)


class MultiHeadAttention(nn.Module):
    """Parameter holder with the reference's names (transformer.py:12-39); runs inside TransformerEncoderLayer."""

    def __init__(self, embed_dim: int, num_heads: int, dropout: float, layer_norm_eps: float):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (embed_dim, num_heads))
        if embed_dim // num_heads > 64 or embed_dim % 4 != 0:
            raise NotImplementedError("HIP path: head size <= 64 and hidden size divisible by 4")
        self.num_attention_heads = num_heads
        self.attention_head_size = embed_dim // num_heads
        self.all_head_size = embed_dim
        self.query = nn.Linear(embed_dim, embed_dim)
        self.key = nn.Linear(embed_dim, embed_dim)
        self.value = nn.Linear(embed_dim, embed_dim)
        self.attn_dropout = nn.Dropout(dropout)
        self.dense = nn.Linear(embed_dim, embed_dim)
        self.LayerNorm = nn.LayerNorm(embed_dim, eps=layer_norm_eps)
        self.out_dropout = nn.Dropout(dropout)


class FeedForward(nn.Module):
    def __init__(self, d_model: int, dim_feedforward: int, dropout: float,
                 activation: str | Callable[[torch.Tensor], torch.Tensor], layer_norm_eps: float, residual: bool = True):
        super().__init__()
        self.dense_1 = nn.Linear(d_model, dim_feedforward)
        self.act_code = _act_code(activation)
        self.dense_2 = nn.Linear(dim_feedforward, d_model)
        self.residual = residual
        if self.residual:                                      # as upstream: created here, used when NOT residual
            self.LayerNorm = nn.LayerNorm(d_model, eps=layer_norm_eps)
            self.dropout = nn.Dropout(dropout)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, dropout: float = 0.1,
                 activation: str | Callable[[torch.Tensor], torch.Tensor] = F.relu, layer_norm_eps: float = 1e-5):
        super().__init__()
        self.multi_head_attention = MultiHeadAttention(d_model, nhead, dropout, layer_norm_eps)
        self.feed_forward = FeedForward(d_model, dim_feedforward, dropout, activation, layer_norm_eps)
        self.dropout_p = float(dropout)
        self.eps = float(layer_norm_eps)

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
        a, f = self.multi_head_attention, self.feed_forward
        if not f.residual:
            # upstream fails here too: a FeedForward built with residual=False has no LayerNorm / dropout to apply
            raise AttributeError("'FeedForward' object has no attribute 'dropout'")
        meta = dict(heads=a.num_attention_heads, dff=f.dense_1.out_features, act=f.act_code, dropout=self.dropout_p,
                    eps=self.eps, training=self.training)
        return _EncoderLayerFn.apply(hidden_states, attention_mask, meta, a.query.weight, a.query.bias, a.key.weight,
                                     a.key.bias, a.value.weight, a.value.bias, a.dense.weight, a.dense.bias,
                                     a.LayerNorm.weight, a.LayerNorm.bias, f.dense_1.weight, f.dense_1.bias,
                                     f.dense_2.weight, f.dense_2.bias)


class TransformerEncoder(nn.Module):
    """``num_layers`` deep copies of one layer (identical initial weights, transformer.py:165-168)."""

    def __init__(self, encoder_layer: nn.Module, num_layers: int):
        super().__init__()
        self.layer = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])

    def forward(self, hidden_states, attention_mask: torch.Tensor, **kwargs):
        for layer_module in self.layer:
            hidden_states = layer_module(hidden_states, attention_mask, **kwargs)
        return hidden_states
