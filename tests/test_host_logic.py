"""Host-side logic that needs no GPU: config schema, flat parameter layout vs the reference's
state-dict names, synthetic batch format, algorithmic FLOP accounting."""
import importlib.util
import os
import sys

import pytest
import torch

from gamer_amd import synthetic
from gamer_amd.config import Qwen3MultiConfig, synthetic_config
from gamer_amd.engine import ParamLayout
from oracle import qwen3multi_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_roundtrip_and_validation(tmp_path):
    cfg = synthetic_config()
    cfg.validate()
    assert cfg.vocab_size == 1041 and "num_positions" in cfg and cfg.num_experts == 6
    cfg.save_pretrained(str(tmp_path))
    cfg2 = Qwen3MultiConfig.from_pretrained(str(tmp_path))
    assert cfg2.to_dict() == cfg.to_dict()
    bad = synthetic_config(head_dim=32)
    with pytest.raises(ValueError):
        bad.validate()
    lut = cfg.behavior_lut()
    assert lut.shape == (1041,) and lut[1038] == 0 and lut[1040] == 2 and int((lut >= 0).sum()) == 3


def test_reference_config_json_loads_when_present():
    path = "/root/reference/config/s2s-models/Qwen3Multi"
    if not os.path.isdir(path):
        pytest.skip("reference not mounted")
    cfg = Qwen3MultiConfig.from_pretrained(path)
    assert cfg.hidden_size == 256 and cfg.cross_attention_decoder == [4, 5, 6, 7] and cfg.vocab_size == 14


def test_hf_config_object_is_accepted_as_the_reference_passes_it(tmp_path):
    """ref:SeqRec/tasks/train_SMB_decoder.py:231, 335-368: the task loads a transformers Qwen3MoeConfig from config.json,
    sets the run-time fields as ATTRIBUTES and hands the object to the model.  Qwen3MultiConfig.coerce (what
    Qwen3MultiWithTemperature.__init__ runs) must read it field by field - rope_theta sits in `rope_parameters` under
    transformers 5.x."""
    transformers = pytest.importorskip("transformers")
    from transformers.models.qwen3_moe import Qwen3MoeConfig
    syn = synthetic_config()
    syn.save_pretrained(str(tmp_path))                      # the reference's config.json schema
    base = {k: v for k, v in syn.to_dict().items()
            if k not in ("num_behavior", "behavior_maps", "use_behavior_token", "num_positions", "num_experts",
                         "n_positions", "use_user_token", "model_max_length", "vocab_size")}
    import json
    json.dump(base, open(tmp_path / "config.json", "w"))
    hf = Qwen3MoeConfig.from_pretrained(str(tmp_path))
    with pytest.raises(ValueError, match="run-time fields"):
        Qwen3MultiConfig.coerce(hf)
    hf.num_behavior, hf.behavior_maps = 3, {"1038": 0, "1039": 1, "1040": 2}
    hf.use_behavior_token, hf.num_positions, hf.num_experts, hf.n_positions = True, 5, 6, 101
    hf.use_user_token, hf.model_max_length, hf.vocab_size = False, 1024, 1041
    cfg = Qwen3MultiConfig.coerce(hf)
    cfg.validate()
    want = syn.to_dict()
    got = cfg.to_dict()
    for k in want:
        if k in ("torch_dtype", "architectures", "rope_scaling", "model_type"):      # bookkeeping fields HF rewrites
            continue
        assert got[k] == want[k], k
    assert float(cfg.rope_theta) == 1e6
    assert Qwen3MultiConfig.coerce(cfg) is cfg and Qwen3MultiConfig.coerce(want).to_dict() == want
    # a config.json written by transformers 5.x (rope_parameters instead of rope_theta) loads too
    hf.save_pretrained(str(tmp_path / "hf"))
    assert float(Qwen3MultiConfig.from_pretrained(str(tmp_path / "hf")).rope_theta) == 1e6


def test_param_layout_matches_reference_state_dict():
    cfg = synthetic_config()
    layout = ParamLayout(cfg)
    assert layout.numel == 24_535_040                      # parameter count of the reference model
    shapes = orc.param_shapes(orc.OracleConfig.from_dict(cfg.to_dict()))
    assert set(layout.entries) == set(shapes)
    for k, (off, shp) in layout.entries.items():
        assert tuple(shp) == tuple(shapes[k]), k
        assert off % 4 == 0
        assert (off >= layout.n_decay) == orc.is_no_decay(k), k
    # fused operands are contiguous: q|k|v rows; gate_e | up_e per expert (one [2 I, din] operand, stride 2 I din between
    # experts), the down weights stacked behind them
    e = layout.entries
    q, k_, v = (e[f"model.layers.5.cross_attn.{n}_proj.weight"][0] for n in "qkv")
    assert k_ - q == 384 * 256 and v - k_ == 192 * 256
    g0 = e["model.layers.0.mlp.experts.expert_0.gate_proj.weight"][0]
    u0 = e["model.layers.0.mlp.experts.expert_0.up_proj.weight"][0]
    g1 = e["model.layers.0.mlp.experts.expert_1.gate_proj.weight"][0]
    assert u0 - g0 == 512 * 320 and g1 - g0 == 2 * 512 * 320
    d0 = e["model.layers.0.mlp.experts.expert_0.down_proj.weight"][0]
    d1 = e["model.layers.0.mlp.experts.expert_1.down_proj.weight"][0]
    assert d0 - g0 == 6 * 2 * 512 * 320 and d1 - d0 == 256 * 512
    flat = torch.arange(layout.numel, dtype=torch.float32)
    views = layout.views(flat)
    assert views["model.norm.weight"].shape == (256,) and float(views["model.embed_tokens.weight"][0, 0]) == 0.0


def test_flat_optimizer_state_of_an_older_parameter_order_is_adopted_by_name():
    """Checkpoints store the Adam moments flat.  Round 4 moved the experts' gate / up weights (gate_e | up_e adjacent for the fused
    projection): a file written before must land on the right parameters, element for element."""
    cfg = synthetic_config(num_hidden_layers=2, behavior_injection_decoder=[0], cross_attention_decoder=[1], codebook=8)
    new, old = ParamLayout(cfg), ParamLayout(cfg, version=1)
    assert new.numel == old.numel and set(new.entries) == set(old.entries) and new.version == ParamLayout.VERSION
    assert new.entries != old.entries                       # the order did change
    g = torch.Generator().manual_seed(0)
    flat_old = torch.randn(old.numel, generator=g)
    named = {k: v.clone() for k, v in old.views(flat_old).items()}
    flat_new = new.adopt(flat_old, cfg, None)               # None: a file from before the layouts were numbered
    for k, v in new.views(flat_new).items():
        assert torch.equal(v, named[k]), k
    assert new.adopt(flat_new, cfg, ParamLayout.VERSION) is flat_new
    with pytest.raises(ValueError):
        new.adopt(flat_old[:-4], cfg, 1)


def test_no_decay_rule_matches_hf_trainer_when_reference_present():
    """oracle.is_no_decay restates HF Trainer.get_decay_parameter_names for this model."""
    if not os.path.isdir("/root/reference/SeqRec"):
        pytest.skip("reference not mounted")
    try:
        from transformers import Trainer
    except Exception:
        pytest.skip("transformers not importable")
    from oracle import _ref_loader
    Model, Cfg = _ref_loader.load_reference_classes()
    cfg = _ref_loader.reference_config(Cfg, 3, 49, synthetic.behavior_maps(8, 3), num_hidden_layers=5,
                                       sparse_layers_decoder=[0, 1, 2, 3, 4], cross_attention_decoder=[4])
    model = Model(cfg)
    decay = set(Trainer.get_decay_parameter_names(None, model))
    for name, _ in model.named_parameters():
        assert (name not in decay) == orc.is_no_decay(name), name


def test_synthetic_batch_format():
    b = synthetic.make_batch(6, 9, 8, 3, ragged=True, seed=4)
    ids, am, act, lab = b["input_ids"], b["attention_mask"], b["actions"], b["labels"]
    B, S = ids.shape
    assert S % 5 == 0 and ids.dtype == torch.int64
    assert bool(((ids == synthetic.PAD_ID) == (am == 0)).all())
    assert bool((act[am == 0] == 100).all()) and bool((act[am == 1] <= 2).all())
    beh = ids[:, ::5]
    assert bool(((beh >= 14 + 32) | (beh == synthetic.PAD_ID)).all())
    assert bool((lab[:, ::5] == -100).all()) and bool((lab[am == 0] == -100).all())
    inner = lab.view(B, -1, 5)[:, :, 1:]
    assert bool((inner[am.view(B, -1, 5)[:, :, 1:] == 1] >= 14).all())
    # right padding by whole items
    lens = am.sum(1)
    assert bool((lens % 5 == 0).all()) and bool((am[:, :1] == 1).all())
    # deterministic
    b2 = synthetic.make_batch(6, 9, 8, 3, ragged=True, seed=4)
    assert all(torch.equal(b[k], b2[k]) for k in b)


def test_algorithmic_flops_match_survey_numbers():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = synthetic_config()
    b = synthetic.make_batch(2, 101, 256, 3, seed=0)
    f = bench.algorithmic_flops(cfg, b)
    assert f["per_token"] == 14_950_912
    assert f["p_self"] == 2 * 127_765
    # brute-force the cross pairs
    a, keep = b["actions"], b["attention_mask"].bool()
    S = a.shape[1]
    i = torch.arange(S).view(1, S, 1)
    j = torch.arange(S).view(1, 1, S)
    ok = (j <= i) & (a[:, None, :] < a[:, :, None]) & keep[:, None, :]
    assert f["p_cross"] == int(ok.sum())
    assert f["step"] == 3 * f["fwd"]


def test_attention_dropout_hash_statistics():
    """numpy restatement of csrc/common.h:AttnDropout (multiply-shift mask: keep(i,j) = low32(a_i*b_j) >= p*2^32).
    Row / key keep rates, adjacent-element correlations and the 2x2 interaction must look like independent
    Bernoulli(0.8) draws at the shapes the model uses (S = 505, 6 heads)."""
    import numpy as np

    def mix32(x):
        x = x.astype(np.uint64) & 0xffffffff
        x ^= x >> 16
        x = (x * 0x7feb352d) & 0xffffffff
        x ^= x >> 15
        x = (x * 0x846ca68b) & 0xffffffff
        x ^= x >> 16
        return x

    p, S, rows = 0.2, 505, 505 * 6
    for seed in (1, 0x1234567890ab, (7 << 48) | (3 << 16) | 0x25):
        k0 = int(mix32(np.array([(seed & 0xffffffff) ^ 0x9e3779b9]))[0]) ^ int(mix32(np.array([((seed >> 32) + 0x7f4a7c15)]))[0])
        k1 = int(mix32(np.array([k0 + 0x632be5ab]))[0])
        a = (mix32(np.arange(rows, dtype=np.uint64) * 0x9e3779b1 + k0) & 0xffffff) | 0x800001
        b = mix32(np.arange(S, dtype=np.uint64) * 0x85ebca6b + k1) & 0xffffff
        keep = ((a[:, None] * b[None, :]) & 0xffffffff) >= int(p * 2 ** 32)
        sd_row, sd_col = np.sqrt(0.16 / S), np.sqrt(0.16 / rows)
        assert abs(keep.mean() - 0.8) < 2e-3
        assert abs(keep.mean(1).std() - sd_row) < 0.15 * sd_row
        assert abs(keep.mean(0).std() - sd_col) < 0.25 * sd_col
        assert abs(np.corrcoef(keep[:, :-1].ravel(), keep[:, 1:].ravel())[0, 1]) < 5e-3
        assert abs(np.corrcoef(keep[:-1].ravel(), keep[1:].ravel())[0, 1]) < 5e-3
        quad = (keep[:-1, :-1] & keep[:-1, 1:] & keep[1:, :-1] & keep[1:, 1:]).mean()
        assert abs(quad - 0.8 ** 4) < 4e-3


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under gamer_amd/ may import it (statically), and importing the
    whole product package must not pull it in (dynamically)."""
    import ast
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "gamer_amd")
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), f"{fn} imports the oracle"
    code = ("import sys, importlib, pkgutil, gamer_amd\n"
            "for m in pkgutil.iter_modules(gamer_amd.__path__):\n"
            "    if m.name not in ('build',): importlib.import_module('gamer_amd.' + m.name)\n"
            "bad = [k for k in sys.modules if k == 'oracle' or k.startswith('oracle.')]\n"
            "assert not bad, bad\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_rccl_log_parser():
    """bench.py --gpus N puts what RCCL decided into its JSON line (gamer_amd.dp.parse_rccl_log): checked here on the line
    shapes NCCL / RCCL print with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH,TUNING."""
    from gamer_amd.dp import parse_rccl_log
    log = """\\
host:101:101 [0] NCCL INFO RCCL version 2.22.3+hip7.2 HEAD:abc
host:101:140 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7
host:101:140 [0] NCCL INFO Channel 15/16 :    0   7   6   5   4   3   2   1
host:101:140 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC
host:101:140 [0] NCCL INFO Channel 01 : 0[0] -> 1[1] via P2P/IPC
host:101:140 [0] NCCL INFO Channel 00 : 0[0] -> 7[7] via P2P/direct pointer
host:101:140 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 nvmlDev 0 busId 5000 commId 0x1 - Init COMPLETE
host:101:101 [0] NCCL INFO 12582912 Bytes -> Algo 1 proto 2 time 210.5
host:101:101 [0] NCCL INFO 12582912 Bytes -> Algo 1 proto 2 time 210.5
host:101:101 [0] NCCL INFO 4 Bytes -> Algo 0 proto 0 time 9.1
"""
    r = parse_rccl_log(log)
    assert r["version"].startswith("2.22.3") and r["nranks"] == 8 and r["channels"] == 16
    assert r["transports"] == {"P2P/IPC": 2, "P2P/direct pointer": 1}
    assert r["algo_proto"]["Ring/Simple"] == {"collectives": 2, "bytes": 2 * 12582912}
    assert r["algo_proto"]["Tree/LL"] == {"collectives": 1, "bytes": 4}
    assert parse_rccl_log("")["algo_proto"] == {} and parse_rccl_log("garbage\\nlines")["nranks"] is None
