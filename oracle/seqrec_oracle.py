"""CPU restatement (test infrastructure: imported by tests/ only) of the `train_decoder` plumbing of BASELINE configs[0]:
leave-last-two-out samples, EncoderDecoderCollator layout and TIGER's temperature loss, in plain Python / numpy.
Pinned by tests/golden/seqrec_small.npz, which oracle/make_golden_seqrec.py generates from the real reference classes."""
import numpy as np

PAD, EOS, IGNORE = 0, 1, -100


def samples(inters: dict, mode: str, max_his_len: int):
    """ref:SeqRec/datasets/seq_dataset.py:95-143 -> list of (history item ids, target item id)."""
    out = []
    for items in inters.values():
        if mode == "train":
            body = items[:-2]
            pairs = [(body[:i], body[i]) for i in range(1, len(body))]
        elif mode == "valid":
            pairs = [(items[:-2], items[-2])]
        else:
            pairs = [(items[:-1], items[-1])]
        for h, t in pairs:
            out.append((h[-max_his_len:] if max_his_len > 0 else h, t))
    return out


def vocabulary(indices: dict, base: int = 32100):
    """seq_dataset.py:26-36 + ref:SeqRec/tasks/train_decoder.py:239-240: add_tokens(sorted(new tokens)) after the T5 vocabulary."""
    return {t: base + i for i, t in enumerate(sorted({t for idx in indices.values() for t in idx}))}


def collate(batch, indices: dict, vocab: dict, model_max_length: int):
    """ref:SeqRec/datasets/collator.py:13-28: both sides get </s>, are truncated to model_max_length and right-padded."""
    def enc(items):
        toks = [vocab[t] for i in items for t in indices[str(i)]]
        return toks[:model_max_length - 1] + [EOS]
    ins = [enc(h) for h, _ in batch]
    outs = [enc([t]) for _, t in batch]
    Li, Lo = max(map(len, ins)), max(map(len, outs))
    ids = np.array([r + [PAD] * (Li - len(r)) for r in ins], dtype=np.int64)
    lab = np.array([r + [PAD] * (Lo - len(r)) for r in outs], dtype=np.int64)
    lab[lab == PAD] = IGNORE
    return ids, (ids != PAD).astype(np.int64), lab


def temperature_ce(logits: np.ndarray, labels: np.ndarray, temperature: float):
    """ref:SeqRec/models/generative/TIGER/model.py:153-161: mean CE of logits / T over labels != -100 (fp64 here).
    Returns (loss, d loss / d logits)."""
    z = logits.astype(np.float64).reshape(-1, logits.shape[-1]) / temperature
    y = labels.reshape(-1)
    keep = y != IGNORE
    m = z.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(z - m).sum(axis=1))
    n = int(keep.sum())
    rows = np.nonzero(keep)[0]
    loss = float((lse[rows] - z[rows, y[rows]]).sum() / n)
    g = np.exp(z - lse[:, None])
    g[rows, y[rows]] -= 1.0
    g[~keep] = 0.0
    return loss, (g / (n * temperature)).reshape(logits.shape)
