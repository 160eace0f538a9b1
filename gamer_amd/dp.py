"""Pure data parallelism for the train step: one process per GPU, gradients summed with RCCL.

The reference's only parallelism is DDP over NCCL (ref:SeqRec/tasks/multi_gpu.py:53,
ref:SeqRec/tasks/train_SMB_decoder.py:420): one bucketed all-reduce of all 24.5 M fp32 gradients per
optimizer step, overlapped with backward.  Here the gradient is one flat fp32 buffer laid out layer
by layer, so bucket l (the decayed weights of decoder layer l, ~12 MB) is a contiguous slice that is
reduced asynchronously the moment layer l's backward has been enqueued; the tail bucket (embedding +
all RMSNorm weights) goes last.  ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for the tests) runs the collective on its own stream; the compute stream only waits in
``finish()``.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def layer_buckets(layout, num_layers: int) -> Tuple[List[Tuple[int, int]], List[Tuple[int, int]]]:
    """(per-layer slices, tail slices) of the flat buffer.  Together they cover [0, numel) exactly once."""
    first = [layout.entries[f"model.layers.{l}.self_attn.q_proj.weight"][0] for l in range(num_layers)]
    emb_end = first[0]
    layers = []
    for l in range(num_layers):
        end = first[l + 1] if l + 1 < num_layers else layout.n_decay
        layers.append((first[l], end))
    tail = [(0, emb_end), (layout.n_decay, layout.numel)]
    return layers, tail


class GradAllReducer:
    """Sums the flat gradient over the process group, bucket by bucket, overlapped with backward."""

    def __init__(self, flat_grad: torch.Tensor, layout, num_layers: int, group=None):
        self.flat = flat_grad
        self.group = group
        self.layers, self.tail = layer_buckets(layout, num_layers)
        self.pending = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # gloo cannot be relied on to take device tensors (it is the CPU backend of the tests; tests/test_dp_gpu.py runs
        # two ranks on ONE GPU with it): device buckets go through a host copy for that backend only.  RCCL ("nccl")
        # reduces the device slices in place.
        self.stage = bool(dist.is_initialized() and dist.get_backend(group) == "gloo" and flat_grad.is_cuda)
        self.staged = []

    def _reduce(self, a: int, b: int):
        if self.stage:
            host = self.flat[a:b].cpu()              # synchronises with the stream that wrote the bucket
            self.staged.append((a, b, host))
            self.pending.append(dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.pending.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def layer_done(self, l: int):
        if self.world == 1:
            return
        self._reduce(*self.layers[l])

    def finish(self):
        if self.world == 1:
            return
        for a, b in self.tail:
            self._reduce(a, b)
        for w in self.pending:
            w.wait()
        for a, b, host in self.staged:
            self.flat[a:b].copy_(host)
        self.pending, self.staged = [], []


def all_reduce_scalar_(t: torch.Tensor, group=None):
    """In-place SUM of a tiny device tensor (the global label count for the loss normalisation,
    HF average_tokens_across_devices semantics)."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("GAMER_BENCH_FORCE_DIST") == "1"):
        if t.is_cuda and dist.get_backend(group) == "gloo":
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def already_reduced_hook(state, bucket):
    """DDP communication hook for a module whose backward has ALREADY reduced its gradients (the nn.Module path after
    ``model.enable_dp_overlap()``: the engine launches its per-layer buckets while the remaining layers' backward kernels
    run, which torch DDP cannot do for a model that is one autograd node).  Returns the bucket untouched:
    ``ddp_model.register_comm_hook(None, gamer_amd.dp.already_reduced_hook)``."""
    fut = torch.futures.Future()
    fut.set_result(bucket.buffer())
    return fut
