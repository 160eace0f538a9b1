"""Evaluation path of the SMB decoder: trie-constrained beam search over the HIP engine.

Mirrors what ``test_SMB_decoder.py:test_single_behavior`` obtains from
``model.generate(..., max_new_tokens=4, num_beams=k, num_return_sequences=k, prefix_allowed_tokens_fn=...,
early_stopping=True)`` (ref:SeqRec/tasks/test_SMB_decoder.py:163-180) with the per-behaviour ``Trie`` of
``prefix_allowed_tokens_fn_by_last_token`` (ref:SeqRec/generation/trie.py:5-104; built at
test_SMB_decoder.py:470-500): the same sequences in the same order and the same ``sequences_scores``.

* The trie lives on the device as a CSR array; ``gamer_trie_logprobs`` does log-softmax + constraint + beam
  score per row and ``gamer_trie_advance`` moves the beams' trie nodes (csrc/decode.hip).  The reference walks a
  Python dict per (sample, beam) per step on the host.
* Beam bookkeeping follows transformers' ``GenerationMixin._beam_search`` for this call: top 2k of k*V
  candidates, the best k continue, the best k of the last step are the hypotheses, score = sum of log-probs /
  number of new tokens.  No EOS can be produced (the trie never allows it).
* K/V cache (``DecodeSession``): the prompt runs once per sample and its keys / values are stored once per sample
  (the reference stores num_beams copies); a generation step pushes one token per beam through the layers and
  ``gamer_attn_decode``.  ``use_cache=False`` re-runs the whole sequence each step instead; both give the same
  beams.  The two things the reference's cache freezes are explicit (``act_zero_col``, ``uniform_len``;
  Engine.forward).  One reference defect is NOT reproduced: its cross-attention cache lives on the module and is not re-ordered with the beams
  (model.py:569,785,844-860), which perturbs samples whose target row is "empty" (oracle/decode_oracle.py,
  tests/test_decode.py quantify it).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch

from . import ops


class ItemTrie:
    """CSR trie over item token sequences (node 0 = root).  ``get(prefix)`` is the host-side ``Trie.get``."""

    def __init__(self, sequences: Sequence[Sequence[int]], device="cuda", pad_token_id: int = 4):
        """``pad_token_id``: the tokenizer's pad id (config.pad_token_id; 4 in the shipped config.json), which
        test_SMB_decoder.py:473-474 adds to the item-ending tokens."""
        nodes: List[Dict[int, int]] = [{}]
        for seq in sequences:
            cur = 0
            for t in seq:
                t = int(t)
                nxt = nodes[cur].get(t)
                if nxt is None:
                    nxt = len(nodes)
                    nodes[cur][t] = nxt
                    nodes.append({})
                cur = nxt
        start, tok, child = [0], [], []
        for d in nodes:
            for t, c in d.items():             # insertion order, as the reference's dict
                tok.append(t)
                child.append(c)
            start.append(len(tok))
        self.nodes = nodes
        self.n_items = len(sequences)
        # tokens that end an item, plus the pad id (test_SMB_decoder.py:473-474)
        self.last_tokens = {int(seq[-1]) for seq in sequences} | {int(pad_token_id)}
        i32 = dict(dtype=torch.int32, device=device)
        self.child_start = torch.tensor(start, **i32)
        self.child_tok = torch.tensor(tok if tok else [0], **i32)
        self.child_node = torch.tensor(child if child else [0], **i32)

    def get(self, prefix: Sequence[int]) -> List[int]:
        cur = 0
        for t in prefix:
            cur = self.nodes[cur].get(int(t))
            if cur is None:
                return []
        return list(self.nodes[cur].keys())


def prefix_allowed_tokens(trie: ItemTrie):
    """Callable with the signature of the reference's ``prefix_allowed_tokens_fn_by_last_token`` result
    (trie.py:90-104) for host-side use; ``Qwen3MultiWithTemperature.generate`` takes the trie it carries."""
    last_tokens = {seq_last for seq_last in trie.last_tokens}

    def fn(batch_id: int, sentence) -> List[int]:
        s = [int(t) for t in sentence]
        i = len(s) - 1
        while i >= 0 and s[i] not in last_tokens:
            i -= 1
        return trie.get(s[i + 1:])
    fn.trie = trie
    return fn


def trie_from_callable(fn, input_ids, max_new_tokens: int, device="cuda", pad_token_id: int = 4) -> ItemTrie:
    """Device trie equivalent to an arbitrary ``prefix_allowed_tokens_fn(batch_id, sentence) -> allowed token ids`` for
    prompts that end with a target behaviour token - the closure the evaluation task builds per behaviour
    (ref:SeqRec/tasks/test_SMB_decoder.py:470-500, ref:SeqRec/generation/trie.py:90-104) passed to ``generate`` unchanged.
    The callable is walked depth first from one prompt per distinct final token (its answers only depend on the tokens
    since the last item-ending token, i.e. on [behaviour token] + generated), ``max_new_tokens`` levels deep: one host
    call per trie node, once - the result is cached on the callable (``fn._gamer_tries``).
    LIMITS (checked where they can be): the callable must not depend on ``batch_id`` or on prompt tokens before the last one -
    the first level is cross-checked on a second row with the same final token when the batch has one, and a callable that
    allows nothing after the prompt raises; a callable with other semantics than trie.py:90-104 needs its own ItemTrie."""
    ids = input_ids.detach().cpu()
    cache = getattr(fn, "_gamer_tries", None)
    if cache is None:
        cache = {}
        try:
            fn._gamer_tries = cache
        except AttributeError:                      # a callable without a __dict__: walk it every time
            pass
    last = ids[:, -1]
    key = (tuple(sorted(set(int(t) for t in last.tolist()))), int(max_new_tokens), str(device))
    if key in cache:
        return cache[key]
    sequences = []
    for tok in key[0]:
        b = int((last == tok).nonzero()[0])
        prompt = ids[b].tolist()

        def walk(prefix):
            if len(prefix) == max_new_tokens:
                sequences.append([tok] + prefix)
                return
            allowed = [int(t) for t in fn(b, torch.tensor(prompt + prefix))]
            if not allowed and prefix:                          # an item shorter than max_new_tokens
                sequences.append([tok] + prefix)
            for t in allowed:
                walk(prefix + [t])
        first = sorted(int(t) for t in fn(b, torch.tensor(prompt)))
        if not first:
            raise ValueError(f"prefix_allowed_tokens_fn allows no token after a prompt ending in {tok}: prompts must end with the "
                             "target behaviour token (ref:SeqRec/tasks/test_SMB_decoder.py:470-500)")
        others = (last == tok).nonzero().flatten().tolist()
        if len(others) > 1:                                     # the same final token in another row must give the same first level
            b2 = int(others[-1])
            if sorted(int(t) for t in fn(b2, ids[b2])) != first:
                raise ValueError("prefix_allowed_tokens_fn depends on more than the tokens since the last item-ending token "
                                 f"(rows {b} and {b2} end in {tok} and get different continuations): build an ItemTrie per row group")
        walk([])
    trie = ItemTrie(sequences, device=device, pad_token_id=pad_token_id)
    cache[key] = trie
    return trie


class _DecodeStatic:
    """Buffers of the cached decode path at FIXED addresses, kept on the engine per (batch, prompt length, beams, new tokens, ...):
    an evaluation run decodes many batches of one shape and reuses them; with GAMER_DECODE_GRAPH=1 the per-token step - ~350
    launches - is captured into one hipGraph per step index once two sessions of the shape have run eagerly (kernel attributes
    set, weight maxima registered) and replayed from then on, bit-identical to the eager step (tests/test_decode.py).
    MEASURED on MI355X / ROCm 7.2 (tools/decode_leg.py): the replay is no faster than the eager step - 2.45 against 2.40 ms per token
    at 16 users x 20 beams, 2.28 / 2.28 at 64, 4.08 / 3.98 at 256: a kernel node costs the ~7 us of dispatch an eager launch costs,
    and at 256 users the step is GPU time anyway (tools/decode_step_kernels.py: gamer_attn_decode 1.73 of 3.6 ms) - so the graph
    is OFF by default; what would shorten the step is FEWER launches, not cheaper ones."""

    def __init__(self):
        self.kp: Dict[Tuple[int, str], torch.Tensor] = {}
        self.vp: Dict[Tuple[int, str], torch.Tensor] = {}
        self.qkvp: Dict[Tuple[int, str], torch.Tensor] = {}     # fp32 engines: the layers' own q|k|v buffers (vp = their v columns)
        self.gen = None
        self.buf = None
        self.small: Dict[str, torch.Tensor] = {}
        self.graphs: Dict[int, "torch.cuda.CUDAGraph"] = {}
        self.sessions = 0            # sessions of this shape that have run so far
        self.sig = None              # state of the engine's maxima cache the graphs were captured against


def _graphs_enabled() -> bool:
    import os
    return os.environ.get("GAMER_DECODE_GRAPH", "0") == "1"


class DecodeSession:
    """K/V cache of one generation run + the single-token forward over it (model.py:118-121, 784-785).

    The prompt is run once per SAMPLE (HF expands it to num_beams copies first) and its keys / values are kept
    once per sample; only the generated positions are per beam.  ``step`` pushes one token per beam through the
    layers with the ordinary row kernels (norms, GEMMs with M = B*num_beams rows, SwiGLU) - all new tokens of a
    step sit at the same position, hence in the same position-routed expert - and ``gamer_attn_decode``.
    GAMER_DECODE_GRAPH=1: from the third session of a shape on the step is a hipGraph replay (``_DecodeStatic``: measured neutral)."""

    def __init__(self, engine, input_ids, attention_mask, actions, num_beams: int, max_new_tokens: int,
                 session_ids=None, extended_session_ids=None, reorder_cross_cache: bool = False):
        """``reorder_cross_cache=False`` (default) is the reference as shipped: its cross-attention K/V cache lives on
        the module (model.py:569, 785, 844-860) and HF's beam search only re-orders ``past_key_values``, so the
        generated positions' cross K/V rows stay in the beam SLOT that wrote them.  They are masked for every query
        except target rows with no lower-level key in the prompt, whose uniform average then reads V rows of other
        beams.  ``True`` re-orders the cross cache with the beams like the self cache (the consistent variant; the
        only one the cache-free re-run path can reproduce).
        ``session_ids`` / ``extended_session_ids`` [B, L0] (the test collator's layout, collator.py:176-195) for a
        "session" engine: the prompt runs with the session masks; the generated tokens then see every kept key in
        the self attention and the prompt's last cross-mask row in the cross attention, exactly as for Qwen3Multi
        (Qwen3SessionMulti/model.py:598-613, 716-728), with RoPE positions last extended id + 1, + 2, ...
        (:969-982)."""
        cfg, dev = engine.cfg, engine.device
        self.eng, self.nb, self.tmax = engine, num_beams, max_new_tokens
        self.reorder_cross_cache = bool(reorder_cross_cache)
        self.B, self.L0 = input_ids.shape
        B, L0, nb = self.B, self.L0, num_beams
        self.N = N = B * nb
        self.session = engine.variant == "session"
        statics = engine.__dict__.setdefault("_decode_static", {})
        key = (B, L0, nb, max_new_tokens, engine.variant, self.reorder_cross_cache, engine.matmul)
        st = statics.get(key)
        if st is None:
            if len(statics) >= 4:                     # (a few shapes at most: the buffers of a shape are ~2.5 GB at 256 users)
                statics.pop(next(iter(statics)))
            st = statics[key] = _DecodeStatic()
        self.st = st
        ids0 = input_ids.to(dev, torch.int64)
        am0 = attention_mask.to(dev, torch.int64)
        act0 = actions.to(dev, torch.int64)
        nq, nkv, dh = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        self.NQ, self.NKV = nq * dh, nkv * dh
        f32 = dict(dtype=torch.float32, device=dev)
        self.kp, self.vp = st.kp, st.vp

        # fp32 engines: the prompt pass writes every layer's q|k|v projection and rotated keys straight into the session's buffers
        # (Engine.forward(kv_dest=...): the V cache is the v columns of the layer's own q|k|v buffer, row stride QKV) - no copy of 2 x
        # 98 MB per layer and kind - and the maxima the three-piece decode attention needs are the ones the prompt pass's own
        # attention used (the producers left them in the engine's maxima cache): no gamer_absmax_f32 pass over the caches either
        import os
        QKV = self.NQ + 2 * self.NKV
        T0 = ids0.shape[0] * ids0.shape[1]
        direct = engine.dtype == "f32" and os.environ.get("GAMER_DECODE_DIRECT_KV", "1") != "0"      # (0: the copying path, A/B and tests)
        want_amax = (engine._amax is not None and engine.matmul == "split3" and os.environ.get("GAMER_DECODE_ATTN_SPLIT", "1") != "0")
        prompt_amax = {}

        def dest(layer, kind):
            kk = (layer, kind)
            if kk not in st.qkvp or st.qkvp[kk].shape[0] != T0:
                st.qkvp[kk] = torch.empty(T0, QKV, **f32)
                st.kp[kk] = torch.empty(T0, self.NKV, **f32)
                st.vp[kk] = st.qkvp[kk][:, self.NQ + self.NKV:]
            return st.qkvp[kk], st.kp[kk]

        def sink(layer, kind, k, v):
            kk = (layer, kind)
            if direct:
                if want_amax:       # (called in front of the layer's attention: the slots the producers of k / v left for it, if they did)
                    sk = engine._amax.peek(k, (1, 0, T0, self.NKV, k.stride(0)))
                    sv = engine._amax.peek(v, (1, 0, T0, self.NKV, v.stride(0)))
                    if sk and sv:
                        prompt_amax[kk] = (sk, sv)
                return
            if kk not in st.kp:
                st.kp[kk] = k.clone()
                st.vp[kk] = v.contiguous().clone()       # v is a column slice of the qkv buffer
            else:
                st.kp[kk].copy_(k)
                st.vp[kk].copy_(v)
        if self.session and (session_ids is None or extended_session_ids is None):
            raise ValueError("a session engine needs session_ids and extended_session_ids")
        skw = {}
        if self.session:
            sid0 = session_ids.to(dev, torch.int64)
            ext0 = extended_session_ids.to(dev, torch.int64)
            skw = dict(session_ids=sid0, extended_session_ids=ext0)
        engine.forward(ids0, am0, act0, train=False, act_zero_col=L0 - 1, uniform_len=L0, kv_sink=sink,
                       kv_dest=dest if direct else None, last_row_logits=True, **skw)
        if self.session:
            engine.check_inputs()
        # last-row logits of every sample: the head ran on B rows, not on the whole prompt
        self.prefill_logits = engine.last_logits_buf

        def keep(name, value):
            """the step's small inputs at fixed addresses (the captured step reads them)"""
            value = value.contiguous()
            t = st.small.get(name)
            if t is None or t.shape != value.shape or t.dtype != value.dtype:
                st.small[name] = t = value.clone()
            else:
                t.copy_(value)
            return t
        # masks of the new rows: self = kept keys; cross = kept keys of a lower level than the target behaviour
        # (the cached last mask row, model.py:603-617); no allowed key -> uniform over every key
        lv = act0[:, -1:]
        self.ok_self = keep("ok_self", am0.to(torch.int32))
        ok_cross = (am0[:, :] != 0) & (act0 < lv)
        self.pos_last = None
        if self.session:
            ok_cross &= sid0 < sid0[:, -1:]              # Qwen3SessionMulti/model.py:582-584, last prompt row
            # RoPE table row of the token generated at step t: the prompt's largest extended id + t
            self.pos_last = keep("pos_last", ext0.max(dim=1).values.to(torch.int32).repeat_interleave(nb))
        ok_cross[:, -1] = False
        self.ok_cross = keep("ok_cross", ok_cross.to(torch.int32))
        self.uniform_cross = keep("uniform_cross", (~ok_cross.any(1)).to(torch.int32))
        # router outputs of the generated tokens (router.py:158-195 in decode mode): behaviour index + 1 of the
        # target item's behaviour token, for the FFN injection and for the cross-attention biases
        beh = engine.lut[ids0[:, -1]].to(torch.int32) + 1
        self.beh = keep("beh", beh.repeat_interleave(nb))
        if st.gen is None:
            st.gen = {kk: (torch.zeros(N, max_new_tokens, self.NKV, **f32), torch.zeros(N, max_new_tokens, self.NKV, **f32))
                      for kk in self.kp}
            H, I = cfg.hidden_size, cfg.intermediate_size
            QKV = self.NQ + 2 * self.NKV
            din_max = H + cfg.behavior_embedding_dim
            st.buf = dict(x=[torch.empty(N, H, **f32) for _ in range(3)], h=torch.empty(N, H, **f32),
                          qkv=torch.empty(N, QKV, **f32), q=torch.empty(N, self.NQ, **f32), k=torch.empty(N, self.NKV, **f32),
                          ao=torch.empty(N, self.NQ, **f32), op=torch.empty(N, H, **f32), gate=torch.empty(N, H, **f32),
                          t0=torch.empty(N, H, **f32), hin=torch.empty(N, din_max, **f32), gu=torch.empty(N, 2 * I, **f32),
                          hm=torch.empty(N, I, **f32), xn=torch.empty(N, H, **f32),
                          logits=torch.empty(N, engine.ws.ldl, **f32),
                          gen_tmp=torch.empty(N, max(1, max_new_tokens - 1), self.NKV, **f32))
            st.small["tok"] = torch.zeros(N, dtype=torch.int64, device=dev)
            st.small["parent"] = torch.arange(N, dtype=torch.int64, device=dev)
        # (stale generated rows of an earlier session are never read: gamer_attn_decode takes the number of valid positions)
        self.gen, self.buf = st.gen, st.buf
        self._pending_reorder = False
        self.t = 0
        # matmul = "split3": the decode attention in the three-piece fp16 form too - the maxima of the prompt K / V caches, which do not
        # change during the generation, measured once (GAMER_DECODE_ATTN_SPLIT=0: the fp32-MFMA kernel)
        self.kv_amax = {}
        if want_amax:
            if len(prompt_amax) == len(self.kp):
                self.kv_amax = prompt_amax
            else:
                with ops.f32_matmul("split3"), engine._amax:
                    for kk in self.kp:
                        kpt, vpt = self.kp[kk], self.vp[kk]
                        self.kv_amax[kk] = (ops.absmax_slot(kpt, 1, 0, kpt.shape[0], kpt.shape[1], kpt.stride(0)),
                                            ops.absmax_slot(vpt, 1, 0, vpt.shape[0], vpt.shape[1], vpt.stride(0)))
        # the maxima cache as this session found it after the prompt pass: what a captured step was recorded against
        am_ = engine._amax
        self._sig = None if am_ is None else (len(am_._wkeys), am_.used, 0 if am_.planes is None else am_.planes.data_ptr())
        if st.graphs and st.sig != self._sig:
            st.graphs.clear()                       # (another weight set / slot layout: record again)
            st.sessions = 0
        st.sessions += 1

    def reorder(self, parent: torch.Tensor):
        """Beams were re-ordered: the generated part of the SELF cache follows its beam (the prompt part is shared);
        the cross cache only with ``reorder_cross_cache`` (see __init__).  The rows move at the start of the next step
        (in place, the positions generated so far only), so that the move is part of the captured step."""
        self.st.small["parent"].copy_(parent)
        self._pending_reorder = True

    @ops.scoped_f32_matmul(lambda self, *a: self.eng.matmul)
    @ops.scoped_amax(lambda self, *a: self.eng._amax)
    def step(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens [N] int64: the token just appended to every beam.  Returns the next-token logits [N, ld].
        (matmul="split3": runs inside the engine's maxima cache - the parameters keep the slots of the prompt pass, which
        measured them once (they cannot change during a generation), the producers of the step's activations hand their
        maxima to the GEMMs (gamer_amax_sink) - instead of two gamer_absmax_f32 launches per GEMM and token.)"""
        st = self.st
        st.small["tok"].copy_(tokens)
        self.t += 1
        t = self.t
        reorder = self._pending_reorder
        self._pending_reorder = False
        if t >= 2 and not reorder:
            st.small["parent"].copy_(torch.arange(self.N, dtype=torch.int64, device=tokens.device))
        g = st.graphs.get(t)
        if g is not None:
            g.replay()
            return self.buf["logits"]
        if _graphs_enabled() and st.sessions >= 3 and tokens.is_cuda:
            self.eng.rope(self.L0 + self.tmax)          # (cached tables: nothing may be built on the host during the capture)
            if self.eng._amax is not None:
                self.eng._amax.reserve(512)             # (a fresh slot pool is zero-filled at allocation: not inside the graph)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_body(t)
            st.graphs[t] = g
            st.sig = self._sig
            g.replay()
            return self.buf["logits"]
        self._step_body(t)
        return self.buf["logits"]

    def _step_body(self, t: int):
        eng, cfg, b = self.eng, self.eng.cfg, self.buf
        N, B, nb, L0 = self.N, self.B, self.nb, self.L0
        H, I, E = cfg.hidden_size, cfg.intermediate_size, cfg.num_experts
        nq, nkv, NQ, NKV = cfg.num_attention_heads, cfg.num_key_value_heads, self.NQ, self.NKV
        QKV = NQ + 2 * NKV
        eps, scale = float(cfg.rms_norm_eps), float(cfg.head_dim) ** -0.5
        tokens, parent = self.st.small["tok"], self.st.small["parent"]
        if t >= 2:
            # the beams were re-ordered after the last step: the t - 1 generated positions follow them
            for key, (kg, vg) in self.gen.items():
                if key[1] == "cross" and not self.reorder_cross_cache:
                    continue
                tmp = b["gen_tmp"].view(-1)[:N * (t - 1) * NKV].view(N, t - 1, NKV)
                torch.index_select(kg[:, :t - 1], 0, parent, out=tmp)
                kg[:, :t - 1] = tmp
                torch.index_select(vg[:, :t - 1], 0, parent, out=tmp)
                vg[:, :t - 1] = tmp
        p = L0 + t - 1                                   # position of the new token
        cos, sin = eng.rope(L0 + self.tmax)
        pos_ids = None
        if self.session:
            pos_ids = (self.pos_last + t).contiguous()   # per beam row; the table is indexed through pos_ids
        else:
            cos, sin = cos[p:p + 1], sin[p:p + 1]
        e = p % cfg.num_positions + 1                    # position-routed expert (router.py:83-104), same for every row
        grp = None
        if eng._amax is not None and eng.matmul == "split3":
            key = f"expert_offsets_{e}"
            if key not in self.st.small:           # rows [0, N) belong to expert e, every other group is empty
                self.st.small[key] = torch.tensor([0] * (e + 1) + [N] * (E - e), dtype=torch.int32, device=tokens.device)
            grp = dict(groups=E, group_offsets=self.st.small[key])
        import os
        fused_ok = eng.dtype == "f32" and os.environ.get("GAMER_DECODE_FUSE_QK", "1") != "0"
        x, x1, x2 = b["x"]
        ops.embedding_fwd(tokens, eng.params["model.embed_tokens.weight"], x)

        # the per-head RMSNorm + RoPE as the q|k|v projection's epilogue: at these row counts the projection runs on the 128 x 128 kernel
        # either way (the activation-stationary kernel starts at 16 k rows), so the epilogue only removes a launch and a pass per
        # attention and token (in the train step, where it would displace the faster kernel, it measured 9 ms slower)
        fuse_qk = fused_ok and ops.qkv_fused_ok(b["h"], N, QKV)

        def attend(kind, layer, Wa, hin, xin, xout, act_idx):
            bias = dict(bias_q=Wa["bq"], bias_k=Wa["bk"], bias_v=Wa["bv"], act_idx=act_idx) if kind == "cross" else {}
            if fuse_qk:
                ops.gemm(hin, H, 1, Wa["qkv"], H, 1, b["qkv"], QKV, N, QKV, H,
                         qknorm=dict(wq=Wa["qn"], wk=Wa["kn"], eps=eps, cos=cos, sin=sin, q_rot=b["q"], k_rot=b["k"], pos_ids=pos_ids,
                                     S=1, nq=nq, nkv=nkv, **bias))
            else:
                ops.linear_fwd(hin, H, Wa["qkv"], H, b["qkv"], QKV, N, QKV, H)
                ops.qknorm_rope_fwd(b["qkv"], 1, nq, nkv, Wa["qn"], Wa["kn"], eps, cos, sin, b["q"], b["k"], pos_ids=pos_ids, **bias)
            kg, vg = self.gen[(layer, kind)]
            ops.kv_append(b["k"], b["qkv"][:, NQ + NKV:], kg, vg, t - 1)
            ops.attn_decode(b["q"], self.kp[(layer, kind)], self.vp[(layer, kind)],
                            self.ok_self if kind == "self" else self.ok_cross, kg, vg, t, kind == "self",
                            None if kind == "self" else self.uniform_cross, B, nb, L0, nq, nkv, scale, b["ao"],
                            amax=self.kv_amax.get((layer, kind)))
            if kind == "self":
                ops.gemm(b["ao"], NQ, 1, Wa["o"], NQ, 1, xout, H, N, H, NQ, resid=xin)
            else:
                ops.linear_fwd(b["ao"], NQ, Wa["o"], NQ, b["op"], H, N, H, NQ)
                ops.linear_fwd(hin, H, Wa["gate"], H, b["gate"], H, N, H, H)
                ops.silu_gate_fwd(b["op"], b["gate"], xout, resid=xin)

        # x holds the layer input and receives the layer output; x1 / x2 are the states after the self / cross block
        for l in range(cfg.num_hidden_layers):
            W = eng.W[l]
            ops.rmsnorm_fwd(x, W.ln1, eps, b["h"])
            attend("self", l, W.self_attn, b["h"], x, x1, None)
            xc = x1
            if W.cross:
                ops.rmsnorm_fwd(x1, W.ln2, eps, b["h"])
                attend("cross", l, W.cross_attn, b["h"], x1, x2, self.beh)
                xc = x2
            din = W.din
            ops.rmsnorm_fwd(xc, W.ln3, eps, b["hin"], din)
            if W.inject:
                ops.rowtable_fwd(W.beh, self.beh, b["hin"], din, H)
            if grp is not None:
                # the whole stacked weight with every row in expert e's group: the tensor the prompt pass measured and cut (its maximum
                # slot and packed pieces are reused; a row slice is another tensor to the maxima cache - two gamer_absmax_f32 launches
                # and two cuts of the weight per layer and token)
                ops.linear_fwd(b["hin"], din, W.gu, din, b["gu"], 2 * I, N, 2 * I, din, strideB=2 * I * din, **grp)
                ops.swiglu_fwd_ld(b["gu"], 2 * I, N, I, 0.0, 0, b["hm"])
                ops.gemm(b["hm"], I, 1, W.down, I, 1, x, H, N, H, I, strideB=H * I, resid=xc, **grp)
                continue
            ops.linear_fwd(b["hin"], din, W.gu[e * 2 * I:(e + 1) * 2 * I], din, b["gu"], 2 * I, N, 2 * I, din)      # gate | up of expert e
            ops.swiglu_fwd_ld(b["gu"], 2 * I, N, I, 0.0, 0, b["hm"])
            ops.gemm(b["hm"], I, 1, W.down[e * H:(e + 1) * H], I, 1, x, H, N, H, I, resid=xc)
        cur = x
        ops.rmsnorm_fwd(cur, eng.params["model.norm.weight"], eps, b["xn"])
        ops.linear_fwd(b["xn"], H, eng.params["model.embed_tokens.weight"], H, b["logits"], b["logits"].stride(0), N,
                       cfg.vocab_size, H)


@torch.no_grad()
def beam_search(engine, input_ids: torch.Tensor, attention_mask: torch.Tensor, actions: torch.Tensor, trie: ItemTrie,
                num_beams: int, max_new_tokens: int = 4, use_cache: bool = True, session_ids=None,
                extended_session_ids=None, reorder_cross_cache: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """input_ids / attention_mask / actions: [B, L0] left-padded prompts ending with the target behaviour token.
    Returns (sequences [B*num_beams, L0+max_new_tokens] int64, sequences_scores [B*num_beams] fp32), the beams of
    sample b at rows b*num_beams .., best first - the layout of HF's GenerateBeamOutput.
    ``use_cache=False`` re-runs the whole sequence every step (the cross-check of the cache path; it has no slots, so
    it needs ``reorder_cross_cache=True``).  ``reorder_cross_cache``: see ``DecodeSession`` (False = the reference).
    ``session_ids`` / ``extended_session_ids`` [B, L0]: required by a "session" engine (see DecodeSession)."""
    if not use_cache and not reorder_cross_cache:
        raise ValueError("use_cache=False re-computes every position for its own beam: it can only reproduce "
                         "reorder_cross_cache=True (the shipped reference's un-reordered cross cache needs the cache)")
    dev = engine.device
    B, L0 = input_ids.shape
    V = engine.cfg.vocab_size
    nb, K = num_beams, 2 * num_beams
    N = B * nb
    ids0 = input_ids.to(dev, torch.int64)
    am0 = attention_mask.to(dev, torch.int64)
    act0 = actions.to(dev, torch.int64)
    seqs = ids0[:, None, :].expand(B, nb, L0).contiguous()
    run_scores = torch.zeros(B, nb, device=dev)
    run_scores[:, 1:] = -1e9                       # only beam 0 is live at the first step (HF)
    # trie node of every beam: the prefix since the last complete item is the target behaviour token
    node = torch.zeros(N, dtype=torch.int32, device=dev)
    nxt = torch.empty_like(node)
    ops.trie_advance(node, ids0[:, -1].repeat_interleave(nb).contiguous(), trie.child_start, trie.child_tok,
                     trie.child_node, nxt)
    node, nxt = nxt, node
    scores = torch.empty(N, V, device=dev)
    final = None
    sess_variant = engine.variant == "session"
    if sess_variant:
        if session_ids is None or extended_session_ids is None:
            raise ValueError("a session engine needs session_ids and extended_session_ids")
        sid0 = session_ids.to(dev, torch.int64)
        ext0 = extended_session_ids.to(dev, torch.int64)

    def session_kw(step, repeat):
        """ids of prompt + ``step`` generated tokens for the full re-run: the generated tokens carry the behaviour
        token's session id and the next extended ids, which makes the training masks equal to the cached ones"""
        if not sess_variant:
            return {}
        sid = torch.cat([sid0, sid0[:, -1:].expand(B, step)], 1)
        ext = torch.cat([ext0, ext0[:, -1:] + torch.arange(1, step + 1, device=dev)[None, :]], 1)
        if repeat:
            sid, ext = sid.repeat_interleave(nb, 0), ext.repeat_interleave(nb, 0)
        return dict(session_ids=sid, extended_session_ids=ext)
    session = (DecodeSession(engine, ids0, am0, act0, nb, max_new_tokens, session_ids if sess_variant else None,
                             extended_session_ids if sess_variant else None, reorder_cross_cache=reorder_cross_cache)
               if use_cache else None)
    last_tok = None
    for step in range(max_new_tokens):
        cur = L0 + step
        if session is not None:
            if step == 0:
                logits2d = session.prefill_logits
                rows = torch.arange(N, device=dev, dtype=torch.int32) // nb
            else:
                logits2d = session.step(last_tok)
                rows = torch.arange(N, device=dev, dtype=torch.int32)
        elif step == 0:
            # all beams of a sample hold the same prompt: run it once (HF runs num_beams copies)
            engine.forward(ids0, am0, act0, train=False, act_zero_col=L0 - 1, uniform_len=L0, **session_kw(0, False))
            rows = (torch.arange(N, device=dev, dtype=torch.int32) // nb) * cur + (cur - 1)
        else:
            flat = seqs.reshape(N, cur)
            am = torch.cat([am0, am0.new_ones(B, step)], 1).repeat_interleave(nb, 0)
            # generated tokens belong to the target item: same behaviour level as its behaviour token
            act = torch.cat([act0, act0[:, -1:].expand(B, step)], 1).repeat_interleave(nb, 0)
            engine.forward(flat, am, act, train=False, act_zero_col=L0 - 1, uniform_len=L0, **session_kw(step, True))
            rows = torch.arange(N, device=dev, dtype=torch.int32) * cur + (cur - 1)
        if session is None:
            logits2d = engine.ws.logits
        ops.trie_logprobs(logits2d, rows, run_scores.reshape(N).contiguous(), node, trie.child_start, trie.child_tok,
                          V, scores)
        top_s, top_i = torch.topk(scores.view(B, nb * V), K)
        beam_i, tok = top_i // V, top_i % V
        cand = torch.cat([torch.gather(seqs, 1, beam_i[:, :, None].expand(B, K, cur)), tok[:, :, None]], 2)
        if step == max_new_tokens - 1:
            final = (cand[:, :nb].reshape(N, cur + 1), (top_s[:, :nb] / max_new_tokens).reshape(N))
            break
        seqs = cand[:, :nb].contiguous()
        run_scores = top_s[:, :nb].contiguous()
        parent = (beam_i[:, :nb] + torch.arange(B, device=dev)[:, None] * nb).reshape(N)
        last_tok = tok[:, :nb].reshape(N).contiguous()
        if session is not None:
            session.reorder(parent)
        ops.trie_advance(node[parent].contiguous(), tok[:, :nb].reshape(N).contiguous(), trie.child_start,
                         trie.child_tok, trie.child_node, nxt)
        node, nxt = nxt.clone(), nxt
    return final
