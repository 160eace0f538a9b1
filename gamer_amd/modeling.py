"""nn.Module surface of the reference model, backed by the HIP engine.

``Qwen3MultiWithTemperature`` mirrors ref:SeqRec/models/generative/Qwen3Multi/model.py:883-1013:
same constructor (a config object with the fields of train_SMB_decoder.py:335-360), same
``set_hyper`` / ``forward`` keyword surface, same state-dict key names (so checkpoints written by
the reference load unchanged), same return fields (``loss``, ``logits``; tuple and key access).
Parameters are views into the engine's flat fp32 buffer; the whole forward/backward is one
``torch.autograd.Function`` so HF ``Trainer`` / DDP can drive it, while ``Engine.train_step`` is the
fused fast path used by bench.py.  There is no CPU implementation here.
"""
from __future__ import annotations

import json
import math
import os
from collections import OrderedDict
from typing import Optional

import torch
from torch import nn

from .config import Qwen3MultiConfig
from .engine import Engine


class CausalLMOutput(OrderedDict):
    """Minimal stand-in for transformers' CausalLMOutputWithPast: attribute, key and index access."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        if isinstance(k, int):
            return [v for v in self.values() if v is not None][k]
        return super().__getitem__(k)


def _autocast_dtype():
    """dtype of the autocast region the caller runs in, or None.  HF Trainer / accelerate wrap ``forward`` in
    ``torch.autocast(device_type, dtype=torch.bfloat16)`` when the task passes ``TrainingArguments(bf16=True)``
    (ref:SeqRec/tasks/train_SMB_decoder.py:114-118, 407-408) - that region IS the reference's bf16 switch."""
    try:
        if torch.is_autocast_enabled("cuda"):
            return torch.get_autocast_dtype("cuda")
    except TypeError:                                         # older torch: no device argument
        if torch.is_autocast_enabled():
            return torch.get_autocast_gpu_dtype()
    return None


try:                                                          # isinstance(model, GenerationMixin) is how the evaluation task decides
    from transformers.generation.utils import GenerationMixin as _GenerationMixin    # whether to call model.generate or
except Exception:                                             # noqa: BLE001          model.module.generate (test_SMB_decoder.py:141-145)
    class _GenerationMixin:                                   # transformers absent: nothing checks the type
        pass


class _ModelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, eng, input_ids, attention_mask, actions, labels, num_items, sess, *params):
        loss, logits = eng.forward(input_ids, attention_mask, actions, labels=labels, num_items_in_batch=num_items,
                                   train=True, dropout=model.training, session_ids=sess[0],
                                   extended_session_ids=sess[1])
        # The engine's logits live in a workspace buffer that this step's backward turns into d(logits) and the next
        # forward overwrites, so the module hands out a COPY by default (custom compute_loss, training-time metrics and
        # label smoothing read outputs.logits after backward).  A training loop that only reads `loss` (HF
        # Trainer.training_step, bench.py --path module) sets `model.zero_copy_logits = True` and gets the view
        # instead of a 2.2 GB copy per forward at batch 1024.
        if not model.zero_copy_logits:
            logits = logits.clone()
        ctx.model, ctx.eng = model, eng
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        eng: Engine = ctx.eng          # the engine the forward ran on (fp32, or the bf16 one under autocast): same flat gradient buffer
        eng.zero_grad()
        red = ctx.model._reducer
        done = red.layer_done if red is not None else None
        if dloss is None:
            eng.backward(0.0, layer_done=done)
        else:
            # the incoming gradient stays on the device (no host read between forward and backward)
            eng.backward(1.0, dloss_dev=dloss.detach().to(torch.float32).reshape(1).contiguous(), layer_done=done)
        if red is not None:
            red.finish()
        # one 98 MB device copy (~40 us): autograd accumulates into .grad tensors of its own, the engine's flat
        # gradient buffer is rewritten by the next backward.  After enable_dp_overlap() the sum over ranks becomes DDP's mean.
        flat = eng.flat_g.clone() if red is None or red.world == 1 else eng.flat_g / float(red.world)
        grads = [flat[o:o + math.prod(s)].view(s) for o, s in (eng.layout.entries[k] for k in ctx.model._param_keys)]
        return (None, None, None, None, None, None, None, None, *grads)


class _Holder(nn.Module):
    """Name-only node of the parameter tree (the computation lives in the engine)."""

    def forward(self, *args, **kwargs):                      # pragma: no cover
        raise RuntimeError("holder modules only carry parameter names; call the top-level module")


class _NormHolder(_Holder):
    """Holder of an RMSNorm weight.  Registered with transformers' ALL_LAYERNORM_LAYERS so that Trainer versions
    that exclude norm weights from weight decay by module TYPE (4.x) do so here as well; 5.x matches by name."""


def _is_norm_key(key: str) -> bool:
    return key.endswith("norm.weight")                       # input_layernorm, post_*_layernorm, q_norm, k_norm, model.norm


try:                                                          # best effort: transformers is optional
    from transformers.pytorch_utils import ALL_LAYERNORM_LAYERS as _ALL_LN
    if _NormHolder not in _ALL_LN:
        _ALL_LN.append(_NormHolder)
except Exception:                                             # noqa: BLE001
    pass


class Qwen3MultiWithTemperature(nn.Module, _GenerationMixin):
    VARIANT = "multi"

    def __init__(self, config, device: str = "cuda", dtype: str = "f32", matmul: Optional[str] = None):
        """``config``: the object the reference constructs its model from - a transformers ``Qwen3MoeConfig`` loaded from
        config.json and mutated by the task (ref:SeqRec/tasks/train_SMB_decoder.py:231, 335-368) - or a
        ``gamer_amd.config.Qwen3MultiConfig`` / dict with the same fields.  ``self.config`` stays the caller's object
        (the task keeps writing to it: ``model.config.use_cache = False``, :442); the engine gets a plain copy.
        ``dtype="bf16"``: what the reference gets from ``--bf16`` (HF Trainer autocast) is a property of the engine
        here - bf16 matrix operands and activations, fp32 parameters / gradients (the nn.Parameters stay fp32).
        ``matmul``: the engine's form of the fp32 matrix products (None = its default "split3"; "split6" = exact bf16 pieces; "f32" = fp32 MFMA)."""
        nn.Module.__init__(self)
        assert hasattr(config, "num_positions") and isinstance(config.num_positions, int), \
            "Config must have 'num_positions' attribute for Qwen3SessionModel."
        assert hasattr(config, "model_max_length") and isinstance(config.model_max_length, int), \
            "Config must have 'model_max_length' attribute for Qwen3SessionModel."
        self.config = config
        self._cfg = Qwen3MultiConfig.coerce(config)
        self.vocab_size = config.vocab_size
        self.temperature = 1.0
        self.zero_copy_logits = False       # True: the training forward returns a view of the engine's logits buffer
        self._reducer = None                # enable_dp_overlap(): the engine's own per-layer gradient all-reduce
        self.engine = Engine(self._cfg, device=device, temperature=1.0, variant=self.VARIANT, dtype=dtype, matmul=matmul)
        self.engine.init_weights(seed=0)
        self._amp_engine: Optional[Engine] = None     # the bf16 step over the SAME masters / gradients, built on first use under autocast
        self._param_keys = list(self.engine.layout.entries.keys())
        self._register_views()

    def _engine_for_call(self) -> Engine:
        """The reference's precision switch is the caller's autocast region (HF Trainer, ``TrainingArguments(bf16=True)``:
        ref:SeqRec/tasks/train_SMB_decoder.py:114-118, 407-408).  Honour it: under ``torch.autocast(dtype=bfloat16)`` the
        call runs the bf16 step (``Engine(dtype="bf16")``: bf16 matrix operands and activations, fp32 masters, gradients,
        residual stream and loss - what autocast does to the reference) on an engine that shares this module's flat
        parameter and gradient buffers; fp16 autocast is refused (the reference's --fp16 run is not built); no autocast =
        the engine the module was constructed with.  Never a silent fp32 run inside a bf16 region."""
        ac = _autocast_dtype()
        if ac is None or self.engine.dtype == "bf16":
            return self.engine
        if ac != torch.bfloat16:
            raise NotImplementedError(f"autocast dtype {ac}: the bf16 (--bf16) and fp32 steps are built, --fp16 is not")
        if self._amp_engine is None:
            self._amp_engine = Engine(self._cfg, device=str(self.engine.device), temperature=self.temperature,
                                      variant=self.VARIANT, dtype="bf16", share_buffers_of=self.engine)
        self._amp_engine.temperature = float(self.temperature)
        return self._amp_engine

    def enable_dp_overlap(self, group=None):
        """Under DistributedDataParallel (HF Trainer wraps the module, ref:SeqRec/tasks/train_SMB_decoder.py:420) the whole
        model is ONE autograd node, so torch's reducer sees every gradient at the same moment and cannot overlap anything.
        After this call the module's backward reduces the engine's flat gradient itself - per-layer buckets launched while
        the remaining layers' backward runs (gamer_amd.dp.GradAllReducer over RCCL), summed and divided by the world size,
        DDP's mean - and DDP must be told not to reduce again:
            ddp_model.register_comm_hook(None, gamer_amd.dp.already_reduced_hook)
        (with HF Trainer: in ``TrainerCallback.on_train_begin``, ``kwargs["model"]`` is the wrapped model)."""
        from .dp import GradAllReducer
        self._reducer = GradAllReducer(self.engine.flat_g, self.engine.layout, self._cfg.num_hidden_layers, group)
        return self

    def fused_optimizer(self, lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                        max_grad_norm: float = 1.0) -> "FusedClipAdamW":
        """torch.optim.Optimizer over this module's parameters that runs clip_grad_norm_(max_grad_norm) + AdamW as
        the engine's two fused sweeps over the flat buffers (HF Trainer: pass it as ``optimizers=(opt, scheduler)`` and
        set ``max_grad_norm=0`` so that the Trainer does not clip a second time)."""
        return FusedClipAdamW(self, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)

    def _register_views(self):
        """nn.Parameters that alias the engine's flat buffer, hung on a tree of empty holder modules with the
        reference's names (``model.layers.0.self_attn.q_proj.weight`` ...).  The tree is what lets code that walks
        ``named_children()`` / ``_parameters`` - HF ``Trainer.get_decay_parameter_names``, DDP, torch optimizers -
        see the same names and the same no-weight-decay norms as with the reference module."""
        self._params_by_key = {}
        for key in self._param_keys:
            p = nn.Parameter(self.engine.params[key], requires_grad=True)
            self._params_by_key[key] = p
            parts = key.split(".")
            node = self
            for i, name in enumerate(parts[:-1]):
                child = node._modules.get(name)
                if child is None:
                    leaf = i == len(parts) - 2
                    child = _NormHolder() if (leaf and _is_norm_key(key)) else _Holder()
                    node.add_module(name, child)
                node = child
            node.register_parameter(parts[-1], p)

    # ---- reference surface -------------------------------------------------------------------
    def set_hyper(self, temperature: float):
        self.temperature = temperature
        self.engine.temperature = float(temperature)

    def resize_token_embeddings(self, new_num_tokens: int):
        if new_num_tokens == self.config.vocab_size:
            return
        old = {k: v.detach().clone() for k, v in self.state_dict().items()}
        self.config.vocab_size = int(new_num_tokens)
        self._cfg.vocab_size = int(new_num_tokens)
        self.vocab_size = int(new_num_tokens)
        self.engine = Engine(self._cfg, device=str(self.engine.device), temperature=self.temperature,
                             variant=self.VARIANT, dtype=self.engine.dtype, matmul=self.engine.matmul)
        self.engine.init_weights(seed=0)
        self._amp_engine = None                               # (rebuilt over the new buffers on the next autocast call)
        if self._reducer is not None:                         # the flat gradient buffer is a new one
            self.enable_dp_overlap(self._reducer.group)
        for name in list(self._modules):                      # drop the old parameter tree
            del self._modules[name]
        self._param_keys = list(self.engine.layout.entries.keys())
        self._register_views()
        with torch.no_grad():
            for k, p in self.engine.params.items():
                if k == "model.embed_tokens.weight":
                    n = min(p.shape[0], old[k].shape[0])
                    p[:n].copy_(old[k][:n])
                else:
                    p.copy_(old[k])

    def state_dict(self, *args, **kwargs):
        sd = OrderedDict((k, self._params_by_key[k].detach()) for k in self._param_keys)
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]          # tied (model.py:888, config.json:34)
        return sd

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if k != "lm_head.weight"}
        unexpected = [k for k in sd if k not in self.engine.layout.entries]
        if strict and unexpected:
            raise KeyError(f"unexpected keys: {unexpected[:5]}")
        with torch.no_grad():
            self.engine.load_state_dict({k: v for k, v in sd.items() if k in self.engine.layout.entries})

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        self._cfg.save_pretrained(path)        # the reference's config.json schema (readable by both sides)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items() if k != "lm_head.weight"}
        try:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(path, "model.safetensors"))
        except ImportError:
            torch.save(sd, os.path.join(path, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, path: str, device: str = "cuda", dtype: str = "f32", matmul: Optional[str] = None):
        cfg = Qwen3MultiConfig.from_pretrained(path)
        model = cls(cfg, device=device, dtype=dtype, matmul=matmul)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(sd)
        return model

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, actions=None, max_new_tokens: int = 4, num_beams: int = 1,
                 num_return_sequences=None, prefix_allowed_tokens_fn=None, trie=None, **kwargs):
        """The call of test_SMB_decoder.py:163-180: trie-constrained beam search, all beams returned best first.
        ``trie`` is a ``gamer_amd.decode.ItemTrie`` (or pass ``prefix_allowed_tokens_fn=decode.prefix_allowed_tokens(trie)``,
        which carries it); arbitrary Python callables are not supported - the constraint runs on the device.
        Returns an object with ``.sequences`` [B*num_beams, L0+max_new_tokens] and ``.sequences_scores``."""
        from . import decode
        if trie is None:
            trie = getattr(prefix_allowed_tokens_fn, "trie", None)
        if trie is None and callable(prefix_allowed_tokens_fn):
            # the reference's own closure (prefix_allowed_tokens_fn_by_last_token, ref:SeqRec/generation/trie.py:90-104): walked
            # once per target behaviour token into a device trie and cached on the callable
            trie = decode.trie_from_callable(prefix_allowed_tokens_fn, input_ids, max_new_tokens,
                                             device=self.engine.device, pad_token_id=self._cfg.pad_token_id)
        if trie is None:
            raise NotImplementedError("generate() needs trie=ItemTrie(...) or prefix_allowed_tokens_fn (constrained beam search "
                                      "of the SMB evaluation)")
        if num_return_sequences not in (None, num_beams):
            raise NotImplementedError("num_return_sequences must equal num_beams (what the evaluation task uses)")
        if attention_mask is None or actions is None:
            raise ValueError("generate() needs attention_mask and actions")
        seqs, scores = decode.beam_search(self.engine, input_ids, attention_mask, actions, trie, num_beams, max_new_tokens,
                                          session_ids=kwargs.get("session_ids"),
                                          extended_session_ids=kwargs.get("extended_session_ids"),
                                          reorder_cross_cache=bool(kwargs.get("reorder_cross_cache", False)))
        return CausalLMOutput(sequences=seqs, sequences_scores=scores)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                cache_position=None, logits_to_keep=0, session_ids=None, extended_session_ids=None, actions=None,
                **kwargs):
        if (input_ids is None) ^ (inputs_embeds is not None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if inputs_embeds is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("gamer_amd implements the training/scoring forward (no KV cache, no inputs_embeds)")
        if actions is None:
            raise ValueError("Qwen3Multi needs `actions` (behaviour level per token) for the cross-attention mask")
        num_items = kwargs.get("num_items_in_batch", None)
        if torch.is_tensor(num_items):
            num_items = float(num_items)
        eng = self._engine_for_call()
        sess = (session_ids, extended_session_ids) if self.VARIANT == "session" else (None, None)
        needs_grad = torch.is_grad_enabled() and labels is not None
        want_hidden = output_hidden_states if output_hidden_states is not None else bool(getattr(self.config, "output_hidden_states", False))
        hidden = [] if want_hidden else None
        if want_hidden and needs_grad:
            raise NotImplementedError("output_hidden_states is served by the scoring forward (torch.no_grad() or no labels): the "
                                      "training step is one autograd node and hands out no differentiable intermediates")
        if needs_grad:
            params = [self._params_by_key[k] for k in self._param_keys]
            loss, logits = _ModelFn.apply(self, eng, input_ids, attention_mask, actions, labels, num_items, sess, *params)
        else:
            with torch.no_grad():
                loss, logits = eng.forward(input_ids, attention_mask, actions, labels=labels,
                                           num_items_in_batch=num_items, train=False, dropout=False,
                                           session_ids=sess[0], extended_session_ids=sess[1], hidden_sink=hidden)
                # the engine's logits live in a workspace buffer that the next forward overwrites; callers keep
                # module outputs across batches (HF Trainer.predict / evaluate with compute_metrics), so hand out a copy
                logits = logits.clone()
        if self.VARIANT == "session":
            eng.check_inputs()          # session ids out of order cannot be expressed as key spans: raise, do not guess
        if isinstance(logits_to_keep, int) and logits_to_keep > 0:
            logits = logits[:, -logits_to_keep:, :]
        return CausalLMOutput(loss=loss, logits=logits, past_key_values=None,
                              hidden_states=tuple(hidden) if hidden is not None else None, attentions=None)


class FusedClipAdamW(torch.optim.Optimizer):
    """clip_grad_norm_ + AdamW of the HF Trainer defaults (ref:SeqRec/tasks/train_SMB_decoder.py:396-428) as the
    engine's fused kernels (gamer_sumsq + gamer_adamw: 28 B per parameter in one sweep) behind the torch optimizer
    interface: ``param_groups[0]["lr"]`` is what schedulers drive; norm weights get no weight decay (the engine's
    layout keeps them behind ``n_decay``), as HF's ``get_decay_parameter_names`` arranges for the reference."""

    def __init__(self, model: "Qwen3MultiWithTemperature", lr, betas, eps, weight_decay, max_grad_norm):
        self.model = model
        super().__init__([p for p in model.parameters()], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                                               max_grad_norm=max_grad_norm))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng, g = self.model.engine, self.param_groups[0]
        # gather the autograd gradients into the flat buffer (one foreach copy; parameters without a gradient count 0)
        views = eng.layout.views(eng.flat_g)
        have = [(views[k], self.model._params_by_key[k].grad) for k in self.model._param_keys]
        dst = [d for d, s in have if s is not None]
        src = [s for d, s in have if s is not None]
        for d, s in have:
            if s is None:
                d.zero_()
        if dst:
            torch._foreach_copy_(dst, src)
        if eng.flat_m is None:
            eng.flat_m, eng.flat_v = torch.zeros_like(eng.flat_p), torch.zeros_like(eng.flat_p)
        eng.opt_step += 1
        from . import torch_ops
        # the dispatcher-registered op (TORCH_LIBRARY(gamer), csrc/torch_ops.cpp): gamer_sumsq + gamer_adamw
        norm = torch_ops.load().fused_adamw_clip(eng.flat_p, eng.flat_g, eng.flat_m, eng.flat_v, eng.layout.n_decay,
                                                 float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"],
                                                 eng.opt_step, float(g["max_grad_norm"] or 0.0), 1.0)   # 0 = no clipping
        eng.grad_norm.copy_(norm)
        return loss

    def state_dict(self):
        eng = self.model.engine
        return {"state": {"m": eng.flat_m, "v": eng.flat_v, "step": eng.opt_step, "layout_version": eng.layout.version},
                "param_groups": self.param_groups}

    def load_state_dict(self, sd):
        eng = self.model.engine
        st = sd["state"]
        if st.get("m") is not None:
            ver = st.get("layout_version")                # (flat moments: re-ordered by name if written under another parameter order)
            eng.flat_m = eng.layout.adopt(st["m"], eng.cfg, ver).to(eng.device).clone()
            eng.flat_v = eng.layout.adopt(st["v"], eng.cfg, ver).to(eng.device).clone()
        eng.opt_step = int(st.get("step", 0))
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})


class Qwen3SessionMultiWithTemperature(Qwen3MultiWithTemperature):
    """ref:SeqRec/models/generative/Qwen3SessionMulti/model.py:870-1017 - Qwen3Multi's parameters and layers with
    session-wise attention masks (a token sees its own item and strictly earlier sessions; the behaviour-level
    attention additionally needs a lower behaviour level) and RoPE positions = ``extended_session_ids``.
    ``forward`` needs ``session_ids`` (the reference asserts the same) and validates their order on the host."""
    VARIANT = "session"
