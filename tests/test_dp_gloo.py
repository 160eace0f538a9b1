"""Data-parallel gradient reduction on CPU: world_size 2, gloo (the same code runs over RCCL on GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gamer_amd.config import synthetic_config
from gamer_amd.dp import GradAllReducer, all_reduce_scalar_, layer_buckets
from gamer_amd.engine import ParamLayout


def test_buckets_cover_flat_buffer_exactly_once():
    cfg = synthetic_config()
    layout = ParamLayout(cfg)
    layers, tail = layer_buckets(layout, cfg.num_hidden_layers)
    cover = torch.zeros(layout.numel, dtype=torch.int32)
    for a, b in layers + tail:
        assert 0 <= a < b <= layout.numel
        cover[a:b] += 1
    assert bool((cover == 1).all())
    assert len(layers) == 8
    # layer l's bucket holds exactly layer l's decayed weights
    for name, (off, shp) in layout.entries.items():
        if name.startswith("model.layers.") and off < layout.n_decay:
            l = int(name.split(".")[2])
            assert layers[l][0] <= off < layers[l][1], name


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, numel_scale, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = synthetic_config(num_hidden_layers=2, behavior_injection_decoder=[0], cross_attention_decoder=[1],
                               hidden_size=64, moe_intermediate_size=64, intermediate_size=64, codebook=8)
        layout = ParamLayout(cfg)
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(layout.numel, generator=g)
        mine = flat.clone()
        red = GradAllReducer(flat, layout, cfg.num_hidden_layers)
        for l in reversed(range(cfg.num_hidden_layers)):       # backward order
            red.layer_done(l)
        red.finish()
        other = torch.randn(layout.numel, generator=torch.Generator().manual_seed(100 + (1 - rank)))
        ok = torch.allclose(flat, mine + other, atol=1e-6)
        cnt = torch.tensor([3.0 + rank])
        all_reduce_scalar_(cnt)
        ok = ok and float(cnt) == 7.0
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_grad_allreduce_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 1, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out[0] and out[1]
