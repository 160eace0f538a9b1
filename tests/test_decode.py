"""Evaluation path: trie, beam search, ranking metrics.

CPU part: the oracle's beam search (oracle/decode_oracle.py) against tests/golden/decode_small.npz, which holds
what the REAL reference's ``generate`` returned (oracle/make_golden_decode.py) - as shipped, and with its
module-level cross-attention cache re-ordered together with the beams.  The product's host logic
(gamer_amd.decode.ItemTrie, gamer_amd.metrics) against the same fixture.
GPU part: gamer_amd.decode.beam_search over the HIP engine against the fixture.
"""
import json
import os

import numpy as np
import pytest
import torch

from gamer_amd import metrics as gm, synthetic
from gamer_amd.decode import ItemTrie
from oracle import decode_oracle as dec, qwen3multi_oracle as orc

FX = os.path.join(os.path.dirname(__file__), "golden", "decode_small.npz")
FX_SESSION = os.path.join(os.path.dirname(__file__), "golden", "decode_session_small.npz")


def _load(path=FX):
    fx = np.load(path)
    meta = json.loads(str(fx["meta_json"]))
    ocfg = orc.OracleConfig.from_dict(meta["config"])
    sd = orc.init_state_dict(ocfg, seed=meta["weight_seed"])
    for k, v in sd.items():
        if v.dim() == 2:
            sd[k] = v * meta["weight_scale"]
    sd["model.embed_tokens.weight"][synthetic.PAD_ID] = 0
    return fx, meta, ocfg, sd


def _case(fx, tb):
    t = lambda k: torch.from_numpy(fx[f"b{tb}_{k}"])
    return t("input_ids"), t("attention_mask"), t("actions")


def _empty_target_rows(am, act):
    """Samples whose target row has no allowed cross-attention key (no kept key of a lower level)."""
    lv = act[:, -1:]
    return ~(((act[:, :-1] < lv) & (am[:, :-1] == 1)).any(1))


def _check(seq, sc, fx, tb, tag, rows, beams, tol):
    rs = torch.from_numpy(fx[f"b{tb}_sequences{tag}"]).view(-1, beams, seq.shape[-1])
    rsc = torch.from_numpy(fx[f"b{tb}_scores{tag}"]).view(-1, beams)
    seq, sc = seq.cpu().view(-1, beams, seq.shape[-1]), sc.cpu().double().view(-1, beams)
    assert rows.any()
    assert torch.equal(seq[rows], rs[rows]), f"behaviour {tb}{tag}: sequences differ"
    err = float((sc[rows] - rsc[rows]).abs().max())
    assert err < tol, f"behaviour {tb}{tag}: score error {err}"
    return err


@pytest.mark.parametrize("tb", [0, 1, 2])
def test_oracle_beam_search_matches_reference_generate(tb):
    fx, meta, ocfg, sd = _load()
    beams, cb = meta["beams"], meta["codebook"]
    ids, am, act = _case(fx, tb)
    trie = dec.ItemTrie(synthetic.item_tokens(torch.from_numpy(fx["catalogue"]), tb, cb).tolist())
    empty = _empty_target_rows(am, act)
    every = torch.ones_like(empty)
    # the reference AS SHIPPED (cross-attention cache not re-ordered with the beams): every sample of every behaviour,
    # including behaviour 0 whose target rows are all "empty"
    seq, sc = dec.beam_search(sd, ocfg, ids, am, act, trie, beams, 4)
    _check(seq, sc, fx, tb, "", every, beams, 2e-5)
    # option: the cross cache re-ordered like past_key_values (the fixture's patched run): every sample
    seq_f, sc_f = dec.beam_search(sd, ocfg, ids, am, act, trie, beams, 4, reorder_cross_cache=True)
    _check(seq_f, sc_f, fx, tb, "_crossfix", every, beams, 2e-5)
    if empty.any():
        # ... and the defect is real: an empty target row moves the reference's own scores by ~1e-2
        rs = torch.from_numpy(fx[f"b{tb}_scores"]).view(-1, beams)
        rf = torch.from_numpy(fx[f"b{tb}_scores_crossfix"]).view(-1, beams)
        assert float((rs[empty] - rf[empty]).abs().max()) > 1e-4
        others = torch.cat([(rs[~empty] - rf[~empty]).abs().flatten(), torch.zeros(1, dtype=rs.dtype)])
        assert float(others.max()) < 1e-6


@pytest.mark.parametrize("tb", [0, 1, 2])
def test_oracle_session_beam_search_matches_reference_generate(tb):
    """Qwen3SessionMultiWithTemperature.generate (fixture from the real class): same rules as above."""
    fx, meta, ocfg, sd = _load(FX_SESSION)
    assert meta["model"] == "Qwen3SessionMultiWithTemperature"
    beams, cb = meta["beams"], meta["codebook"]
    ids, am, act = _case(fx, tb)
    sess, ext = torch.from_numpy(fx[f"b{tb}_session_ids"]), torch.from_numpy(fx[f"b{tb}_extended_session_ids"])
    assert bool((sess[:, :-1].max(1).values + 1 == sess[:, -1]).all())          # the test collator's layout
    trie = dec.ItemTrie(synthetic.item_tokens(torch.from_numpy(fx["catalogue"]), tb, cb).tolist())
    seq, sc = dec.beam_search(sd, ocfg, ids, am, act, trie, beams, 4, session_ids=sess, extended_session_ids=ext)
    every = torch.ones(ids.shape[0], dtype=torch.bool)
    _check(seq, sc, fx, tb, "", every, beams, 2e-5)                     # the reference as shipped, every row
    seq_f, sc_f = dec.beam_search(sd, ocfg, ids, am, act, trie, beams, 4, session_ids=sess, extended_session_ids=ext,
                                  reorder_cross_cache=True)
    _check(seq_f, sc_f, fx, tb, "_crossfix", every, beams, 2e-5)
    # and the session masks matter on this fixture: Qwen3Multi's masks give other scores
    _, sc_multi = dec.beam_search(sd, ocfg, ids, am, act, trie, beams, 4)
    assert float((sc_multi - sc).abs().max()) > 1e-3


def test_metrics_match_reference_ranking():
    fx, meta, _, _ = _load()
    beams, cb = meta["beams"], meta["codebook"]
    for tb in range(meta["num_behavior"]):
        seqs = torch.from_numpy(fx[f"b{tb}_sequences"])
        scores = torch.from_numpy(fx[f"b{tb}_scores"])
        tgt = synthetic.item_tokens(torch.from_numpy(fx[f"b{tb}_targets"]), tb, cb)[:, 1:]
        targets = [[row.tolist()] for row in tgt]
        topk = gm.get_topk_results(seqs[:, -4:].tolist(), scores.tolist(), targets, beams)
        assert np.array_equal(np.array(topk, dtype=np.int8), fx[f"b{tb}_topk"])
        res = gm.get_metrics_results(topk, meta["metrics"], targets)
        np.testing.assert_allclose([res[m] for m in meta["metrics"]], fx[f"b{tb}_metrics"], rtol=0, atol=1e-12)
    # several targets per sample (recall / ndcg normalisation, ranking.py:36-62)
    topk = [[0, 1, 0, 1, 0], [0, 0, 0, 0, 0]]
    targets = [[[1], [2], [3]], [[4]]]
    res = gm.get_metrics_results(topk, ["hit@5", "recall@5", "ndcg@5"], targets)
    assert res["hit@5"] == 1.0 and abs(res["recall@5"] - 2 / 3) < 1e-12
    ideal = sum(1 / np.log2(j + 2) for j in range(3))
    assert abs(res["ndcg@5"] - (1 / np.log2(3) + 1 / np.log2(5)) / ideal) < 1e-12


def test_item_trie_host_logic():
    cat = synthetic.make_catalogue(40, 8, seed=5)
    items = synthetic.item_tokens(cat, 1, 8).tolist()
    trie, ref = ItemTrie(items, device="cpu"), dec.ItemTrie(items)
    g = torch.Generator().manual_seed(0)
    for _ in range(200):
        it = items[int(torch.randint(0, len(items), (1,), generator=g))]
        n = int(torch.randint(0, 6, (1,), generator=g))
        prefix = it[:n]
        if int(torch.randint(0, 4, (1,), generator=g)) == 0 and n > 0:
            prefix = prefix[:-1] + [3]                       # a token no item has there
        assert trie.get(prefix) == ref.get(prefix)
    # CSR consistency: the children of every node are exactly its dict
    cs, ct, cn = trie.child_start.tolist(), trie.child_tok.tolist(), trie.child_node.tolist()
    for x, d in enumerate(trie.nodes):
        assert dict(zip(ct[cs[x]:cs[x + 1]], cn[cs[x]:cs[x + 1]])) == d
    assert trie.get(items[0]) == []                          # a complete item has no continuation


# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_trie_kernels_against_host():
    from gamer_amd import ops
    dev = "cuda"
    cat = synthetic.make_catalogue(60, 8, seed=9)
    items = synthetic.item_tokens(cat, 2, 8).tolist()
    trie = ItemTrie(items, device=dev)
    V, N = synthetic.vocab_size(8, 3), 37
    g = torch.Generator().manual_seed(1)
    prefixes = []
    for n in range(N):
        it = items[int(torch.randint(0, len(items), (1,), generator=g))]
        prefixes.append(it[:int(torch.randint(0, 5, (1,), generator=g))])
    node = torch.zeros(N, dtype=torch.int32, device=dev)
    for depth in range(4):                                   # walk every row to its prefix with the kernel
        tok = torch.tensor([p[depth] if depth < len(p) else -1 for p in prefixes], device=dev)
        nxt = torch.empty_like(node)
        ops.trie_advance(node, tok, trie.child_start, trie.child_tok, trie.child_node, nxt)
        live = torch.tensor([depth < len(p) for p in prefixes], device=dev)
        node = torch.where(live, nxt, node)
    node[5] = -1                                             # unknown prefix: nothing allowed
    ld = 64
    logits = torch.randn(3 * N, ld, generator=g).to(dev) * 3
    rows = torch.randperm(3 * N, generator=g)[:N].to(dev, torch.int32)
    beam = torch.randn(N, generator=g).to(dev)
    out = torch.empty(N, V, device=dev)
    ops.trie_logprobs(logits, rows, beam, node, trie.child_start, trie.child_tok, V, out)
    ref = torch.full((N, V), float("-inf"), dtype=torch.float64)
    lp = torch.log_softmax(logits.cpu().double()[rows.cpu().long(), :V], -1) + beam.cpu().double()[:, None]
    for n in range(N):
        if n != 5:
            al = trie.get(prefixes[n])
            ref[n, al] = lp[n, al]
    got = out.cpu().double()
    assert torch.equal(torch.isinf(got), torch.isinf(ref))
    fin = ~torch.isinf(ref)
    assert float((got[fin] - ref[fin]).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tb", [0, 1, 2])
def test_engine_beam_search_matches_reference_generate(tb):
    from gamer_amd.config import Qwen3MultiConfig
    from gamer_amd.decode import beam_search
    from gamer_amd.engine import Engine
    fx, meta, ocfg, sd = _load()
    beams, cb = meta["beams"], meta["codebook"]
    cfg = Qwen3MultiConfig(**meta["config"])
    eng = Engine(cfg, temperature=0.7)
    eng.load_state_dict(sd)
    ids, am, act = _case(fx, tb)
    trie = ItemTrie(synthetic.item_tokens(torch.from_numpy(fx["catalogue"]), tb, cb).tolist())
    every = torch.ones(ids.shape[0], dtype=torch.bool)
    otrie = dec.ItemTrie(synthetic.item_tokens(torch.from_numpy(fx["catalogue"]), tb, cb).tolist())
    # default = the reference AS SHIPPED (cross-attention cache not re-ordered with the beams): every row, every
    # behaviour - including behaviour 0, whose target rows are all "empty" - against the un-patched generate()
    seq, sc = beam_search(eng, ids, am, act, trie, beams, 4)
    _check(seq, sc, fx, tb, "", every, beams, 1e-4)
    oseq, osc = dec.beam_search(sd, ocfg, ids, am, act, otrie, beams, 4)
    assert torch.equal(seq.cpu(), oseq) and float((sc.cpu() - osc).abs().max()) < 1e-4
    # option: cross cache re-ordered like the self cache = the patched reference run of the fixture; the K/V-cache
    # path and the re-run-everything path give the same beams
    seq_f, sc_f = beam_search(eng, ids, am, act, trie, beams, 4, reorder_cross_cache=True)
    seq_nc, sc_nc = beam_search(eng, ids, am, act, trie, beams, 4, use_cache=False, reorder_cross_cache=True)
    assert torch.equal(seq_f, seq_nc) and float((sc_f - sc_nc).abs().max()) < 2e-5
    _check(seq_f, sc_f, fx, tb, "_crossfix", every, beams, 1e-4)
    oseq, osc = dec.beam_search(sd, ocfg, ids, am, act, otrie, beams, 4, reorder_cross_cache=True)
    assert torch.equal(seq_f.cpu(), oseq) and float((sc_f.cpu() - osc).abs().max()) < 1e-4
    with pytest.raises(ValueError):
        beam_search(eng, ids, am, act, trie, beams, 4, use_cache=False)


@pytest.mark.gpu
@pytest.mark.parametrize("tb", [0, 1, 2])
def test_engine_session_beam_search_matches_reference_generate(tb):
    """Qwen3SessionMulti: the session engine's cached generation against the real class's ``generate``, the full
    re-run path and the oracle; and through the nn.Module's ``generate``."""
    from gamer_amd.config import Qwen3MultiConfig
    from gamer_amd.decode import beam_search, prefix_allowed_tokens
    from gamer_amd.modeling import Qwen3SessionMultiWithTemperature
    fx, meta, ocfg, sd = _load(FX_SESSION)
    beams, cb = meta["beams"], meta["codebook"]
    model = Qwen3SessionMultiWithTemperature(Qwen3MultiConfig(**meta["config"]))
    model.set_hyper(0.7)
    model.load_state_dict(sd)
    eng = model.engine
    ids, am, act = _case(fx, tb)
    sess, ext = torch.from_numpy(fx[f"b{tb}_session_ids"]), torch.from_numpy(fx[f"b{tb}_extended_session_ids"])
    items = synthetic.item_tokens(torch.from_numpy(fx["catalogue"]), tb, cb).tolist()
    trie = ItemTrie(items)
    every = torch.ones(ids.shape[0], dtype=torch.bool)
    seq, sc = beam_search(eng, ids, am, act, trie, beams, 4, session_ids=sess, extended_session_ids=ext)
    _check(seq, sc, fx, tb, "", every, beams, 1e-4)                     # the reference as shipped, every row
    oseq, osc = dec.beam_search(sd, ocfg, ids, am, act, dec.ItemTrie(items), beams, 4, session_ids=sess,
                                extended_session_ids=ext)
    assert torch.equal(seq.cpu(), oseq) and float((sc.cpu() - osc).abs().max()) < 1e-4
    seq_f, sc_f = beam_search(eng, ids, am, act, trie, beams, 4, session_ids=sess, extended_session_ids=ext,
                              reorder_cross_cache=True)
    seq_nc, sc_nc = beam_search(eng, ids, am, act, trie, beams, 4, use_cache=False, session_ids=sess,
                                extended_session_ids=ext, reorder_cross_cache=True)
    assert torch.equal(seq_f, seq_nc) and float((sc_f - sc_nc).abs().max()) < 2e-5
    _check(seq_f, sc_f, fx, tb, "_crossfix", every, beams, 1e-4)
    oseq, osc = dec.beam_search(sd, ocfg, ids, am, act, dec.ItemTrie(items), beams, 4, session_ids=sess,
                                extended_session_ids=ext, reorder_cross_cache=True)
    assert torch.equal(seq_f.cpu(), oseq) and float((sc_f.cpu() - osc).abs().max()) < 1e-4
    out = model.generate(input_ids=ids, attention_mask=am, actions=act, session_ids=sess, extended_session_ids=ext,
                         max_new_tokens=4, prefix_allowed_tokens_fn=prefix_allowed_tokens(trie), num_beams=beams,
                         num_return_sequences=beams)
    assert torch.equal(out.sequences, seq) and torch.equal(out.sequences_scores, sc)
    with pytest.raises(ValueError):
        beam_search(eng, ids, am, act, trie, beams, 4)                  # the session ids are mandatory


@pytest.mark.gpu
def test_evaluate_behavior_matches_reference_metrics():
    """Behaviour 2 of the fixture: no empty target row, so the reference's own metrics apply unchanged."""
    from gamer_amd.config import Qwen3MultiConfig
    from gamer_amd.engine import Engine
    from gamer_amd.evaluate import evaluate_behavior
    fx, meta, ocfg, sd = _load()
    tb, beams, cb = 2, meta["beams"], meta["codebook"]
    eng = Engine(Qwen3MultiConfig(**meta["config"]), temperature=0.7)
    eng.load_state_dict(sd)
    ids, am, act = _case(fx, tb)
    assert not _empty_target_rows(am, act).any()
    tgt = synthetic.item_tokens(torch.from_numpy(fx[f"b{tb}_targets"]), tb, cb)[:, 1:]
    batch = dict(input_ids=ids, attention_mask=am, actions=act, targets=[[row.tolist()] for row in tgt])
    trie = ItemTrie(synthetic.item_tokens(torch.from_numpy(fx["catalogue"]), tb, cb).tolist())
    # two batches (the same one twice): sums and sample counts accumulate
    res = evaluate_behavior(eng, [batch, batch], trie, beams, meta["metrics"])
    assert res["samples"] == 2 * ids.shape[0]
    np.testing.assert_allclose([res[m] for m in meta["metrics"]], fx[f"b{tb}_metrics"] / ids.shape[0], atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("nq,nkv,nb,L0,t", [(2, 1, 6, 31, 1), (6, 3, 20, 130, 3), (2, 2, 5, 64, 4), (6, 3, 1, 130, 0), (2, 1, 1, 40, 0)])
def test_attn_decode_kernel_against_dense(nq, nkv, nb, L0, t, split):
    """gamer_attn_decode (fp32 MFMA) and gamer_attn_decode_split (three fp16 piece products, the default engine's) vs fp64 softmax
    attention: self (generated keys attended) and cross (generated keys masked, samples without an allowed key uniform over all
    L0 + t keys); the prompt values span four orders of magnitude between rows in the split case (one scale per tensor).
    t = 0: no generated position - a prompt row against the prompt keys (the last layer's cross block in a generation's prompt pass)."""
    from gamer_amd import ops
    dev, B, tmax = "cuda", 3, 4
    N, G = B * nb, nq // nkv
    g = torch.Generator().manual_seed(L0 + t)
    q = torch.randn(N, nq * 64, generator=g)
    kp, vp = torch.randn(B * L0, nkv * 64, generator=g), torch.randn(B * L0, nkv * 64, generator=g)
    kg, vg = torch.randn(N, tmax, nkv * 64, generator=g), torch.randn(N, tmax, nkv * 64, generator=g)
    ok = (torch.rand(B, L0, generator=g) < 0.6).to(torch.int32)
    ok[0, :5] = 0                                            # left padding
    ok[1] = 0                                                # sample 1: nothing allowed
    ok_self = ok.clone()
    ok_self[1, -3:] = 1                                      # self attention always has kept keys
    uniform = (ok.sum(1) == 0).to(torch.int32)
    if split:
        vp = vp * torch.exp(2.3 * torch.randn(B * L0, 1, generator=g))
    vbuf = torch.zeros(B * L0, (nq + 2 * nkv) * 64)          # v lives inside a wider buffer in the engine
    vbuf[:, (nq + nkv) * 64:] = vp
    kp_d, v_d = kp.to(dev), vbuf.to(dev)[:, (nq + nkv) * 64:]
    amax = None
    if split:
        amax = (ops.absmax_slot(kp_d, 1, 0, B * L0, nkv * 64, nkv * 64), ops.absmax_slot(v_d, 1, 0, B * L0, nkv * 64, v_d.stride(0)))
    for kind in ("self", "cross"):
        o = torch.empty(N, nq * 64, device=dev)
        kok = ok_self if kind == "self" else ok
        ops.attn_decode(q.to(dev), kp_d, v_d, kok.to(dev), kg.to(dev), vg.to(dev), t,
                        kind == "self", None if kind == "self" else uniform.to(dev), B, nb, L0, nq, nkv, 0.125, o, amax=amax)
        ref = torch.empty(N, nq, 64, dtype=torch.float64)
        for n in range(N):
            b = n // nb
            for hd in range(nq):
                kv = hd // G
                keys = torch.cat([kp[b * L0:(b + 1) * L0, kv * 64:(kv + 1) * 64], kg[n, :t, kv * 64:(kv + 1) * 64]]).double()
                vals = torch.cat([vp[b * L0:(b + 1) * L0, kv * 64:(kv + 1) * 64], vg[n, :t, kv * 64:(kv + 1) * 64]]).double()
                allowed = torch.cat([kok[b].bool(), torch.full((t,), kind == "self")])
                if kind == "cross" and uniform[b]:
                    ref[n, hd] = vals.mean(0)
                    continue
                s = (keys @ q[n, hd * 64:(hd + 1) * 64].double()) * 0.125
                s[~allowed] = float("-inf")
                ref[n, hd] = torch.softmax(s, 0) @ vals
        err = float((o.cpu().double().view(N, nq, 64) - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (kind, err)


@pytest.mark.gpu
def test_prompt_pass_shortcuts_agree_with_the_plain_pass(monkeypatch):
    """The prompt pass of a generation as built (K / V caches written in place by the layers, their maxima taken from the pass's own
    attention, the last layer's cross block on the last positions only) against the plain one (copies, measured maxima, the whole
    block): same caches bit for bit, prompt logits and the first cached step's logits to fp32 rounding."""
    from gamer_amd.config import synthetic_config
    from gamer_amd.decode import DecodeSession
    from gamer_amd.engine import Engine
    cfg = synthetic_config()
    eng = Engine(cfg, temperature=0.7)
    eng.init_weights(seed=3)
    cat = synthetic.make_catalogue(300, 256, seed=3)
    batch = synthetic.make_eval_batch(5, 30, cat, 2, 256, 3, min_his=12, seed=9, behavior_probs=[0.6, 0.3, 0.1])
    nb = 4
    tok = torch.randint(14, 14 + 256, (5 * nb,), device="cuda")
    out = {}
    for form in ("built", "plain"):
        if form == "plain":
            monkeypatch.setenv("GAMER_DECODE_DIRECT_KV", "0")
            monkeypatch.setenv("GAMER_PREFILL_LAST_ROW", "0")
        eng._decode_static = {}                      # (a fresh set of session buffers for either form)
        s = DecodeSession(eng, batch["input_ids"], batch["attention_mask"], batch["actions"], nb, 3)
        pre = s.prefill_logits.clone()
        caches = {kk: (s.kp[kk].clone(), s.vp[kk].clone()) for kk in s.kp}
        step = s.step(tok).clone()
        torch.cuda.synchronize()
        out[form] = (pre, caches, step)
    V = cfg.vocab_size
    for kk in out["built"][1]:
        assert torch.equal(out["built"][1][kk][0], out["plain"][1][kk][0]), kk
        assert torch.equal(out["built"][1][kk][1], out["plain"][1][kk][1]), kk
    for i in (0, 2):
        a, b = out["built"][i][:, :V].double(), out["plain"][i][:, :V].double()
        assert float((a - b).abs().max()) < 2e-5 * float(b.abs().max()), i


# ---- BASELINE configs[4] at the shipped architecture ------------------------------------------------------------------
def _shipped_case(B, tb, seed, n_catalogue=400):
    from gamer_amd.config import synthetic_config
    cfg = synthetic_config()                                  # hidden 256, 8 layers, 6 / 3 heads, V = 1041 (codebook 256)
    ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
    sd = orc.init_state_dict(ocfg, seed=17)
    for k, v in sd.items():
        if v.dim() == 2:
            sd[k] = v * 4.0                                   # (as the decode fixtures: scores far from uniform, no near-ties)
    sd["model.embed_tokens.weight"][synthetic.PAD_ID] = 0
    cat = synthetic.make_catalogue(n_catalogue, 256, seed=3)
    batch = synthetic.make_eval_batch(B, 100, cat, tb, codebook=256, min_his=70, seed=seed)
    items = synthetic.item_tokens(cat, tb, 256).tolist()
    return cfg, ocfg, sd, batch, items


def _shipped_fixture(sd, tb, ids, am, act):
    """tests/golden/decode_shipped.npz: the reference's generate at the shipped architecture (2 users, 20 beams, his 100) on
    _shipped_case(2, tb, 11 + tb); checks that the inputs and the seeded weights here are the ones the fixture was made from."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "decode_shipped.npz"))
    assert np.array_equal(z[f"b{tb}_input_ids"], ids.numpy()) and np.array_equal(z[f"b{tb}_attention_mask"], am.numpy())
    assert np.array_equal(z[f"b{tb}_actions"], act.numpy())
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - float(z["weight_checksum"][0])) <= 1e-9 * chk, "seeded weights differ from the fixture's"
    return {"sequences": torch.from_numpy(z[f"b{tb}_sequences"]), "scores": torch.from_numpy(z[f"b{tb}_scores"])}


@pytest.mark.parametrize("tb", [0, 2])
def test_decode_oracle_reproduces_the_reference_at_the_shipped_architecture(tb):
    """The CPU decode oracle against the reference's own generate at the shipped dims (20 beams, prompts of 501 tokens, one left
    padded; behaviour 0: every target row "empty", the un-reordered cross cache matters; behaviour 2: none)."""
    cfg, ocfg, sd, batch, items = _shipped_case(2, tb, seed=11 + tb)
    ids, am, act = batch["input_ids"], batch["attention_mask"], batch["actions"]
    fx = _shipped_fixture(sd, tb, ids, am, act)
    oseq, osc = dec.beam_search(sd, ocfg, ids, am, act, dec.ItemTrie(items), 20, 4)
    assert torch.equal(oseq, fx["sequences"]), "oracle beams differ from the reference's generate"
    assert float((osc.double() - fx["scores"]).abs().max()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tb", [0, 2])
def test_beam_search_at_the_shipped_architecture(tb):
    """What test_SMB_decoder.py:158-177 asks of the model at the shipped dims: hidden 256, 8 layers (4 with behaviour
    injection, 4 with the behaviour-level attention), 6 query / 3 kv heads, V = 1041, max_his_len 100 (prompts of 501 tokens,
    left padded), 20 beams, 4 new tokens - the cached generation of gamer_amd.decode against the oracle's beam search run on
    the host (every step re-runs all 20 beams of every sample through oracle.forward).  Behaviour 0: every target row is an
    "empty" row (uniform attention over all keys, the reference's un-reordered cross cache); behaviour 2: none is."""
    from gamer_amd.decode import beam_search
    from gamer_amd.engine import Engine
    B, beams = 2, 20
    cfg, ocfg, sd, batch, items = _shipped_case(B, tb, seed=11 + tb)
    ids, am, act = batch["input_ids"], batch["attention_mask"], batch["actions"]
    assert ids.shape[1] == 501 and int(am[1].sum()) < 501                     # sample 1 is left padded
    eng = Engine(cfg, temperature=0.7)
    eng.load_state_dict(sd)
    seq, sc = beam_search(eng, ids, am, act, ItemTrie(items), beams, 4)
    # pinned DIRECTLY by the reference: tests/golden/decode_shipped.npz holds what the real Qwen3MultiWithTemperature.generate
    # returned on these inputs and weights (oracle/make_golden_decode.py shipped)
    fx = _shipped_fixture(sd, tb, ids, am, act)
    assert torch.equal(seq.cpu(), fx["sequences"]), "beams differ from the reference's generate"
    err_ref = float((sc.cpu().double() - fx["scores"]).abs().max())
    assert err_ref < 1e-4, err_ref
    oseq, osc = dec.beam_search(sd, ocfg, ids, am, act, dec.ItemTrie(items), beams, 4)
    assert torch.equal(seq.cpu(), oseq), "beams differ from the oracle's"
    err = float((sc.cpu() - osc).abs().max())
    assert err < 1e-4, err
    assert len({tuple(r) for r in seq[:beams, -4:].tolist()}) == beams      # 20 distinct items per sample
    # the fp32-MFMA form and the default (three fp16 piece products) find the same beams
    eng32 = Engine(cfg, temperature=0.7, matmul="f32")
    eng32.load_state_dict(sd)
    seq32, sc32 = beam_search(eng32, ids, am, act, ItemTrie(items), beams, 4)
    assert torch.equal(seq32, seq) and float((sc32 - sc).abs().max()) < 1e-4


@pytest.mark.gpu
def test_beam_search_batch_256_rows_equal_rows_alone():
    """configs[4]'s batch: 256 users x 20 beams at the shipped dims.  Size-independent property: the beams of a user inside
    the 256-row batch are the beams of that user decoded alone (same sequences, scores to 2e-5), for rows with the longest,
    a padded and the shortest history; every returned item is a catalogue item of the target behaviour."""
    from gamer_amd.decode import beam_search
    from gamer_amd.engine import Engine
    B, beams, tb = 256, 20, 1
    cfg, ocfg, sd, batch, items = _shipped_case(B, tb, seed=5, n_catalogue=2000)
    ids, am, act = batch["input_ids"], batch["attention_mask"], batch["actions"]
    eng = Engine(cfg, temperature=0.7)
    eng.load_state_dict(sd)
    trie = ItemTrie(items)
    seq, sc = beam_search(eng, ids, am, act, trie, beams, 4)
    assert seq.shape == (B * beams, 505) and bool(torch.isfinite(sc).all())
    lens = am.sum(1)
    rows = sorted({0, int(lens.argmin()), 100, 255})
    sub = torch.tensor(rows)
    seq_s, sc_s = beam_search(eng, ids[sub], am[sub], act[sub], trie, beams, 4)
    for j, r in enumerate(rows):
        assert torch.equal(seq_s[j * beams:(j + 1) * beams], seq[r * beams:(r + 1) * beams]), r
        assert float((sc_s[j * beams:(j + 1) * beams] - sc[r * beams:(r + 1) * beams]).abs().max()) < 2e-5
    valid = {tuple(it[1:]) for it in items}
    assert all(tuple(t) in valid for t in seq[:, -4:].tolist())
    best = sc.view(B, beams)
    assert bool((best[:, :-1] >= best[:, 1:]).all())                          # best first


@pytest.mark.gpu
def test_generate_takes_the_reference_callable_and_is_a_generation_mixin():
    """The evaluation task calls ``(model if isinstance(model, GenerationMixin) else model.module).generate(...,
    prefix_allowed_tokens_fn=<closure over the reference's Trie>, output_scores=True, return_dict_in_generate=True,
    early_stopping=True)`` (ref:SeqRec/tasks/test_SMB_decoder.py:158-177): the module is a GenerationMixin, and an
    arbitrary callable with the reference's semantics (trie.py:90-104) gives the beams of the device trie."""
    transformers = pytest.importorskip("transformers")
    from gamer_amd.config import Qwen3MultiConfig
    from gamer_amd.decode import beam_search
    from gamer_amd.modeling import Qwen3MultiWithTemperature
    fx, meta, ocfg, sd = _load()
    tb, beams, cb = 1, meta["beams"], meta["codebook"]
    model = Qwen3MultiWithTemperature(Qwen3MultiConfig(**meta["config"]))
    model.set_hyper(0.7)
    model.load_state_dict(sd)
    assert isinstance(model, transformers.generation.utils.GenerationMixin)
    ids, am, act = _case(fx, tb)
    items = synthetic.item_tokens(torch.from_numpy(fx["catalogue"]), tb, cb).tolist()
    otrie = dec.ItemTrie(items)
    last_tokens = {it[-1] for it in items} | {synthetic.PAD_ID}
    calls = []

    def by_last_token(batch_id, sentence):                   # the reference's closure, restated over the oracle's trie
        s = sentence.tolist()
        calls.append(len(s))
        i = len(s) - 1
        while i >= 0 and s[i] not in last_tokens:
            i -= 1
        return otrie.get(s[i + 1:])
    out = model.generate(input_ids=ids, attention_mask=am, actions=act, max_new_tokens=4, prefix_allowed_tokens_fn=by_last_token,
                         num_beams=beams, num_return_sequences=beams, output_scores=True, return_dict_in_generate=True,
                         early_stopping=True)
    seq, sc = beam_search(model.engine, ids, am, act, ItemTrie(items), beams, 4)
    assert torch.equal(out.sequences, seq) and torch.equal(out.sequences_scores, sc)
    n_calls = len(calls)
    model.generate(input_ids=ids, attention_mask=am, actions=act, max_new_tokens=4, prefix_allowed_tokens_fn=by_last_token,
                   num_beams=beams, num_return_sequences=beams)
    assert len(calls) == n_calls                             # the walk is cached on the callable


@pytest.mark.gpu
@pytest.mark.parametrize("reorder_cross", [False, True])
def test_decode_step_graph_replay_equals_the_eager_step(reorder_cross, monkeypatch):
    """From the third session of a shape on the cached per-token step is a hipGraph replay over buffers at fixed addresses
    (gamer_amd/decode.py: _DecodeStatic).  Replayed sessions - on the prompts the graphs were captured with AND on other prompts of
    the same shape - return the bits of the eager step (GAMER_DECODE_GRAPH=0), and still the reference's beams."""
    from gamer_amd.config import Qwen3MultiConfig
    from gamer_amd.decode import beam_search
    from gamer_amd.engine import Engine
    fx, meta, ocfg, sd = _load()
    beams, cb = meta["beams"], meta["codebook"]
    cfg = Qwen3MultiConfig(**meta["config"])
    tb = 2
    ids, am, act = _case(fx, tb)
    trie = ItemTrie(synthetic.item_tokens(torch.from_numpy(fx["catalogue"]), tb, cb).tolist())
    # a second batch of the same shape: the users in another order
    rev = torch.arange(ids.shape[0] - 1, -1, -1)
    ids2, am2, act2 = ids[rev], am[rev], act[rev]

    def engine():
        e = Engine(cfg, temperature=0.7)
        e.load_state_dict(sd)
        return e
    monkeypatch.setenv("GAMER_DECODE_GRAPH", "0")
    eng0 = engine()
    ref1 = beam_search(eng0, ids, am, act, trie, beams, 4, reorder_cross_cache=reorder_cross)
    ref2 = beam_search(eng0, ids2, am2, act2, trie, beams, 4, reorder_cross_cache=reorder_cross)
    assert not eng0._decode_static[next(iter(eng0._decode_static))].graphs
    monkeypatch.setenv("GAMER_DECODE_GRAPH", "1")
    eng = engine()
    outs = [beam_search(eng, ids, am, act, trie, beams, 4, reorder_cross_cache=reorder_cross) for _ in range(4)]
    st = eng._decode_static[next(iter(eng._decode_static))]
    assert sorted(st.graphs) == [1, 2, 3]                       # captured in the third session, replayed in the fourth
    for seq, sc in outs:
        assert torch.equal(seq, ref1[0]) and torch.equal(sc, ref1[1])
    seq2, sc2 = beam_search(eng, ids2, am2, act2, trie, beams, 4, reorder_cross_cache=reorder_cross)
    assert torch.equal(seq2, ref2[0]) and torch.equal(sc2, ref2[1])
    # an engine forward of another shape in between (the maxima cache is reset and refilled): the graphs stay valid or are dropped and
    # recorded again - either way the beams are the eager ones
    eng.forward(ids[:2], am[:2], act[:2], train=False)
    seq3, sc3 = beam_search(eng, ids, am, act, trie, beams, 4, reorder_cross_cache=reorder_cross)
    assert torch.equal(seq3, ref1[0]) and torch.equal(sc3, ref1[1])
    if not reorder_cross:
        _check(seq3, sc3, fx, tb, "", torch.ones(ids.shape[0], dtype=torch.bool), beams, 1e-4)
