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


# ---- diagnostics for the multi-GPU runs (bench.py --gpus N): what the collective actually did -----------------------------------
_ALGO = {"0": "Tree", "1": "Ring", "2": "CollNetDirect", "3": "CollNetChain", "4": "NVLS", "5": "NVLSTree", "6": "PAT"}
_PROTO = {"0": "LL", "1": "LL128", "2": "Simple"}


def parse_rccl_log(text: str) -> dict:
    """Summary of an ``NCCL_DEBUG=INFO`` log (RCCL prints the same lines): library version, ranks of the communicator, channel
    count, transports of the peer connections ("via P2P/IPC", "via P2P/direct pointer", "via NET/..."), and - with
    ``NCCL_DEBUG_SUBSYS`` including TUNING - which algorithm / protocol the tuner chose for how many collectives and bytes
    ("<bytes> Bytes -> Algo <a> proto <p> time <t>").  Unknown lines are ignored; an empty log gives empty fields."""
    import re
    from collections import Counter
    out = {"version": None, "nranks": None, "channels": None, "transports": {}, "algo_proto": {}, "lines": 0}
    if not text:
        return out
    out["lines"] = text.count("\n") + 1
    m = re.search(r"(?:RCCL|NCCL) version[ :]+([0-9][0-9A-Za-z.+_-]*)", text)
    if m:
        out["version"] = m.group(1)
    m = re.findall(r"nranks\s+(\d+)", text)
    if m:
        out["nranks"] = int(m[-1])
    ch = [int(b) for _, b in re.findall(r"Channel\s+(\d+)/(\d+)\s*:", text)]
    if ch:
        out["channels"] = max(ch)
    out["transports"] = dict(Counter(re.findall(r"\bvia\s+([A-Za-z0-9/_ -]+?)(?:\s*$|\s+comm|\s*\n)", text, flags=re.M)))
    ap = {}
    for nbytes, algo, proto in re.findall(r"(\d+)\s+Bytes\s*->\s*Algo\s+(\S+)\s+proto\s+(\S+)", text):
        key = f"{_ALGO.get(algo, algo)}/{_PROTO.get(proto, proto)}"
        d = ap.setdefault(key, {"collectives": 0, "bytes": 0})
        d["collectives"] += 1
        d["bytes"] += int(nbytes)
    out["algo_proto"] = ap
    return out


def bucket_timings(reducer: "GradAllReducer", reps: int = 5) -> list:
    """Every gradient bucket of the train step all-reduced ALONE (nothing to overlap with), ``reps`` times: per bucket its
    bytes, mean ms and bus GB/s (2 (N - 1) / N x bytes / time, the ring-equivalent convention).  Run outside the timed region."""
    out = []
    world = max(reducer.world, 1)
    for name, (a, b) in ([(f"layer{l}", ab) for l, ab in enumerate(reducer.layers)] +
                         [("embedding", reducer.tail[0]), ("norms", reducer.tail[1])]):
        if b <= a:
            continue
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            reducer._reduce(a, b)
            for w in reducer.pending:
                w.wait()                         # (the current stream waits for the collective: the end event sees it)
            for aa, bb, host in reducer.staged:
                reducer.flat[aa:bb].copy_(host)
            reducer.pending, reducer.staged = [], []
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        nbytes = (b - a) * 4
        out.append({"bucket": name, "bytes": nbytes, "ms": ms,
                    "bus_GBps": 2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None})
    return out
