#!/usr/bin/env python3
"""Golden fixture for the discriminative baselines' encoder, generated from the REAL
reference classes (build container only): ``TransformerEncoder(TransformerEncoderLayer(...), num_layers)`` of
ref:SeqRec/modules/layers/transformer.py with a SASRec-shaped configuration (hidden 64, 2 heads, inner 256, gelu,
50 positions, additive 0 / -10000 causal + padding mask), dropout 0.  Stores the state dict, inputs, the output and
the gradient of every parameter and of the input.

Usage:  python oracle/make_golden_modules.py        (writes tests/golden/modules_small.npz)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_loader  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "modules_small.npz")
CASES = {"sasrec": dict(D=64, heads=2, dff=256, act="gelu", eps=1e-12, layers=2, B=3, S=50, seed=5),
         "wide": dict(D=128, heads=2, dff=192, act="relu", eps=1e-5, layers=1, B=2, S=37, seed=6)}


def main():
    sys.path.insert(0, _ref_loader.REF_ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_transformer", os.path.join(_ref_loader.REF_ROOT, "SeqRec", "modules", "layers", "transformer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = {}
    for name, c in CASES.items():
        torch.manual_seed(c["seed"])
        layer = mod.TransformerEncoderLayer(c["D"], c["heads"], c["dff"], dropout=0.0, activation=c["act"],
                                            layer_norm_eps=c["eps"])
        enc = mod.TransformerEncoder(layer, c["layers"])
        with torch.no_grad():                      # deep copies start identical: perturb every parameter
            for p in enc.parameters():
                p.add_(0.05 * torch.randn_like(p))
        x = torch.randn(c["B"], c["S"], c["D"], requires_grad=True)
        lens = torch.randint(c["S"] // 3, c["S"] + 1, (c["B"],))
        lens[0] = c["S"]
        keep = torch.arange(c["S"])[None, :] >= (c["S"] - lens)[:, None]                # left padded, as SASRec feeds
        causal = torch.tril(torch.ones(c["S"], c["S"], dtype=torch.bool))
        allowed = causal[None, None] & keep[:, None, None, :]
        mask = torch.where(allowed, 0.0, -10000.0)                                      # [B,1,S,S] additive
        out = enc(x, mask)
        # cotangent on the real positions only (as a loss would): a padded query row sees no allowed key, its
        # scores are all "s - 10000" rounded to fp32's 1e-3 grid at that magnitude, so its softmax - and anything
        # differentiated through it - is decided by rounding noise and is not a meaningful parity target
        w = torch.randn_like(out) * keep[:, :, None]
        (out * w).sum().backward()
        sd = enc.state_dict()
        for k, v in sd.items():
            res[f"{name}/sd/{k}"] = v.numpy()
        for k, p in enc.named_parameters():
            res[f"{name}/grad/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        res[f"{name}/has_grad"] = np.array([k for k, p in enc.named_parameters() if p.grad is not None])
        res.update({f"{name}/x": x.detach().numpy(), f"{name}/mask": mask.numpy(), f"{name}/out": out.detach().numpy(),
                    f"{name}/w": w.numpy(), f"{name}/dx": x.grad.numpy(),
                    f"{name}/keep": keep.numpy()})
        print(name, "out abs max", float(out.abs().max()), "params without grad:",
              [k for k, p in enc.named_parameters() if p.grad is None])
    res["meta_json"] = np.array(json.dumps(dict(cases=CASES, generator=dict(torch=torch.__version__,
                                                                           reference="wzf2000/GAMER @ /root/reference"))))
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
