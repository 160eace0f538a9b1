"""CPU restatement of the reference's post-LN encoder (TEST INFRASTRUCTURE ONLY - never imported by gamer_amd/).

ref:SeqRec/modules/layers/transformer.py:12-183 as plain functions over a state dict with the reference's key
names (``layer.{l}.multi_head_attention.query.weight`` ...):
  attention (:47-82)   q/k/v = Linear(x); scores = q k^T * sqrt(1/head) + additive mask; softmax; dropout;
                       context -> dense -> dropout -> LayerNorm(h + x)
  feed forward (:111-120) dense_2(act(dense_1(x))); the dropout + LayerNorm(h + x) tail sits behind
                       ``if not self.residual`` and is never reached by a layer built with the default residual=True
Pinned by tests/golden/modules_small.npz, generated from the real classes by oracle/make_golden_modules.py.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

ACTS = {"gelu": F.gelu, "relu": F.relu, "swish": F.silu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "elu": F.elu}


def layer_forward(sd: Dict[str, torch.Tensor], prefix: str, x, mask, heads: int, act: str, eps: float):
    B, S, D = x.shape
    dh = D // heads
    a = prefix + "multi_head_attention."
    lin = lambda t, name: F.linear(t, sd[name + ".weight"], sd[name + ".bias"])
    split = lambda t: t.view(B, S, heads, dh).permute(0, 2, 1, 3)
    q, k, v = split(lin(x, a + "query")), split(lin(x, a + "key")), split(lin(x, a + "value"))
    s = torch.matmul(q, k.transpose(-1, -2)) * math.sqrt(1.0 / float(dh))
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    c = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(B, S, D)
    h = lin(c, a + "dense")
    y = F.layer_norm(h + x, (D,), sd[a + "LayerNorm.weight"], sd[a + "LayerNorm.bias"], eps)
    f = prefix + "feed_forward."
    return lin(ACTS[act](lin(y, f + "dense_1")), f + "dense_2")


def encoder_forward(sd, x, mask, num_layers: int, heads: int, act: str, eps: float):
    for l in range(num_layers):
        x = layer_forward(sd, f"layer.{l}.", x, mask, heads, act, eps)
    return x
