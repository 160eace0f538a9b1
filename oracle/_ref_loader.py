"""Loader for the *reference* GAMER model (test infrastructure, build container only).

This file is part of the oracle tooling: it imports the reference's Python from
``/root/reference`` (read-only mount that exists only in the build container) so that
``oracle/make_golden.py`` can generate the committed fixtures under ``tests/golden/``.
Nothing here is shipped or imported by the product package; nothing from the reference is
copied.  The shims below only make the reference importable under python 3.10 /
transformers 5.x.
"""
import importlib.machinery
import os
import sys
import types
import typing

REF_ROOT = os.environ.get("GAMER_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "SeqRec", "models", "generative", "Qwen3Multi"))


def _install_shims():
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    # (1) typing.Unpack on python 3.10
    if not hasattr(typing, "Unpack"):
        import typing_extensions
        typing.Unpack = typing_extensions.Unpack
    # (2) loguru stub
    if "loguru" not in sys.modules:
        class _Logger:
            def __getattr__(self, name):
                if name == "catch":
                    def catch(*a, **k):
                        if len(a) == 1 and callable(a[0]) and not k:
                            return a[0]
                        return lambda f: f
                    return catch
                return lambda *a, **k: None
        stub = types.ModuleType("loguru")
        stub.logger = _Logger()
        sys.modules["loguru"] = stub
    # (3)+(4) names removed from transformers 5.x
    import transformers.models.qwen3.modeling_qwen3 as mq
    if not hasattr(mq, "KwargsForCausalLM"):
        from transformers.modeling_flash_attention_utils import FlashAttentionKwargs
        mq.KwargsForCausalLM = FlashAttentionKwargs
    if not hasattr(mq, "QWEN3_INPUTS_DOCSTRING"):
        mq.QWEN3_INPUTS_DOCSTRING = ""
    # (5) skip Qwen3Moe/__init__ (imports symbols removed in 5.x) but keep Qwen3Moe.FFN importable
    name = "SeqRec.models.generative.Qwen3Moe"
    if name not in sys.modules:
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REF_ROOT, "SeqRec", "models", "generative", "Qwen3Moe")]
        pkg.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        pkg.__spec__.submodule_search_locations = pkg.__path__
        sys.modules[name] = pkg


def load_reference_classes(session: bool = False):
    """Returns (Qwen3MultiWithTemperature, Qwen3MoeConfig) from the reference; ``session=True`` returns
    Qwen3SessionMultiWithTemperature (train_SMB_decoder.py:365-367) instead."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    _install_shims()
    # generative/__init__ may import everything; import the leaf module directly
    for parent in ("SeqRec", "SeqRec.models", "SeqRec.models.generative"):
        if parent not in sys.modules:
            pkg = types.ModuleType(parent)
            pkg.__path__ = [os.path.join(REF_ROOT, *parent.split("."))]
            pkg.__spec__ = importlib.machinery.ModuleSpec(parent, None, is_package=True)
            pkg.__spec__.submodule_search_locations = pkg.__path__
            sys.modules[parent] = pkg
    from transformers.models.qwen3_moe import Qwen3MoeConfig
    if session:
        from SeqRec.models.generative.Qwen3SessionMulti.model import Qwen3SessionMultiWithTemperature
        return Qwen3SessionMultiWithTemperature, Qwen3MoeConfig
    from SeqRec.models.generative.Qwen3Multi.model import Qwen3MultiWithTemperature
    return Qwen3MultiWithTemperature, Qwen3MoeConfig


def reference_config(Qwen3MoeConfig, num_behavior: int, vocab_size: int, behavior_maps: dict,
                     n_positions: int = 101, **overrides):
    """config.json from the reference + the runtime mutation of train_SMB_decoder.py:335-360."""
    cfg = Qwen3MoeConfig.from_pretrained(os.path.join(REF_ROOT, "config", "s2s-models", "Qwen3Multi"))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    cfg.num_behavior = num_behavior
    cfg.behavior_maps = {str(k): int(v) for k, v in behavior_maps.items()}
    cfg.use_behavior_token = True
    cfg.num_positions = 5
    cfg.num_experts = 6
    cfg.n_positions = n_positions
    cfg.use_user_token = False
    cfg.model_max_length = 1024
    cfg.vocab_size = vocab_size
    return cfg
