#!/usr/bin/env python3
"""Host-side data path: collated batches per second of gamer_amd.data against the reference's
SMBExplicitDatasetForDecoder + Qwen2Tokenizer + DecoderOnlyCollator on the same synthetic dataset directory
(the reference leg only runs where /root/reference exists, i.e. in the build container; CPU only).

  python tools/bench_data.py [--users 3000] [--batch 1024] [--max_his_len 100]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=3000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--max_his_len", type=int, default=100)
    ap.add_argument("--augment", type=int, default=4)
    a = ap.parse_args()
    from gamer_amd import data as gdata, synthetic
    tmp = tempfile.mkdtemp(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
    try:
        synthetic.write_smb_dataset(tmp, "Syn", n_users=a.users, n_items=5000, codebook=256, max_sessions=40,
                                    max_per_session=6, seed=2)
        t0 = time.perf_counter()
        ds = gdata.SMBData(tmp, "Syn")
        ss = ds.train_samples(a.max_his_len, augment=a.augment)
        t_build = time.perf_counter() - t0
        coll = gdata.Collator(ds)
        chunks = gdata.batches(len(ss), a.batch, shuffle_seed=0)
        t0 = time.perf_counter()
        ntok = 0
        for idx in chunks:
            b = coll.train(ss, idx)
            ntok += int(b["attention_mask"].sum())
        t_coll = time.perf_counter() - t0
        res = dict(samples=len(ss), tokens=ntok, build_s=t_build, collate_s=t_coll,
                   collate_samples_per_s=len(ss) / t_coll, config=vars(a))
        from oracle import _ref_loader
        if _ref_loader.reference_available():
            _ref_loader.load_reference_classes()
            from SeqRec.datasets.SMB_dataset import SMBExplicitDatasetForDecoder
            from SeqRec.datasets.collator import DecoderOnlyCollator
            from transformers import Qwen2Tokenizer
            t0 = time.perf_counter()
            ref = SMBExplicitDatasetForDecoder(dataset="Syn", data_path=tmp, max_his_len=a.max_his_len,
                                               index_file=".index.json", mode="train", behavior_first=True, augment=a.augment)
            t_rbuild = time.perf_counter() - t0
            tdir = os.path.join(tmp, "_tok")
            os.makedirs(tdir)
            src = os.path.join(_ref_loader.REF_ROOT, "config", "s2s-models", "Qwen3Multi")
            for fn in ("vocab.json", "tokenizer_config.json"):
                shutil.copy(os.path.join(src, fn), os.path.join(tdir, fn))
            open(os.path.join(tdir, "merges.txt"), "w").write("#version: 0.2\n")      # see oracle/make_golden_data.py
            tok = Qwen2Tokenizer.from_pretrained(tdir, model_max_length=1024)
            tok.add_tokens(ref.get_new_tokens())
            beh = [tok.encode(t, add_special_tokens=False)[0] for b_ in ref.behaviors for t in ref.get_behavior_tokens(b_)]
            rc = DecoderOnlyCollator(tok, only_train_response=False, ignore_behavior_tokens=beh)
            n_ref = min(len(ref), 4 * a.batch)
            t0 = time.perf_counter()
            for b0 in range(0, n_ref, a.batch):
                rc([ref[i] for i in range(b0, min(n_ref, b0 + a.batch))])
            t_rcoll = time.perf_counter() - t0
            res["reference"] = dict(samples=len(ref), build_s=t_rbuild, collate_s=t_rcoll, collated=n_ref,
                                    collate_samples_per_s=n_ref / t_rcoll)
        print(json.dumps(res))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
