"""Minimal training harness for the SMB decoder on MI355X: the counterpart of
``TrainSMBDecoder.invoke`` (ref:SeqRec/tasks/train_SMB_decoder.py:139-449) for the hot path only.

Same step semantics as the reference's HF-Trainer run (gradient accumulation, loss = sum CE / number
of label tokens in the accumulation window across ranks, clip_grad_norm_(1.0), AdamW, cosine schedule
with warm-up, per-epoch checkpoints with the reference's file layout), flag names reused from the
reference where they exist.  One process per GPU, launch with torch.distributed.run for more than one.

Data: with ``--data_path/--dataset`` a dataset directory in the reference's on-disk format is read by
``gamer_amd.data`` (``--tasks smb_explicit_decoder[_N]`` as upstream; the reference's own data files are git-LFS
pointers, ``gamer_amd.synthetic.write_smb_dataset`` writes a stand-in in the same format); without it the batches
are synthetic tensors of the same layout.

  python -m gamer_amd.train --max_his_len 100 --per_device_batch_size 128 --gradient_accumulation_steps 4 \
      --epochs 1 --steps_per_epoch 20 --output_dir /tmp/ckpt
  python -m gamer_amd.train --data_path ./data --dataset ShortVideoAD --tasks smb_explicit_decoder_4 \
      --backbone Qwen3Multi --max_his_len 100 --per_device_batch_size 128 --epochs 2
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch

from . import synthetic
from .config import synthetic_config
from .dp import GradAllReducer, all_reduce_scalar_
from .engine import Engine
from .schedule import cosine_with_warmup, warmup_steps_for


def parse_args(argv=None):
    ap = argparse.ArgumentParser("train_SMB_decoder (MI355X hot path)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--max_his_len", type=int, default=100)
    ap.add_argument("--per_device_batch_size", type=int, default=128)
    ap.add_argument("--gradient_accumulation_steps", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--steps_per_epoch", type=int, default=10, help="optimizer steps per synthetic epoch")
    ap.add_argument("--learning_rate", type=float, default=5e-4)
    ap.add_argument("--weight_decay", type=float, default=0.01)
    ap.add_argument("--warmup_ratio", type=float, default=0.1)
    ap.add_argument("--temperature", type=float, default=0.7)
    ap.add_argument("--max_grad_norm", type=float, default=1.0)
    ap.add_argument("--logging_step", type=int, default=30)
    ap.add_argument("--ragged", action="store_true", help="variable history lengths (right padded)")
    ap.add_argument("--data_path", type=str, default="", help="directory holding <dataset>/<dataset>.SMB.*.json")
    ap.add_argument("--dataset", type=str, default="")
    ap.add_argument("--index_file", type=str, default=".index.json")
    ap.add_argument("--tasks", type=str, default="smb_explicit_decoder_4")
    ap.add_argument("--backbone", type=str, default="Qwen3Multi", choices=["Qwen3Multi", "Qwen3SessionMulti"])
    ap.add_argument("--patience", type=int, default=10, help="early stopping: evaluations without a better eval_loss")
    ap.add_argument("--save_total_limit", type=int, default=2)
    ap.add_argument("--bf16", action="store_true",
                    help="the reference's --bf16 (train_SMB_decoder.py:114-118): bf16 matrix operands / activations, "
                         "fp32 master weights, gradients and optimizer state")
    ap.add_argument("--fp16", action="store_true", help="accepted for flag compatibility and refused: not built")
    ap.add_argument("--matmul", choices=["f32", "split3", "split6", "split9"], default=None,
                    help="fp32 runs: how matrix products are formed (default: the engine's, split3 = Linear sites as three fp16 piece "
                         "products scaled per tensor; split6 = exact bf16 piece products with fp32 accumulation; f32 = the fp32 MFMA)")
    ap.add_argument("--prefetch", type=int, default=2, help="optimizer steps collated ahead on a background thread")
    ap.add_argument("--output_dir", type=str, default="")
    ap.add_argument("--resume_from_checkpoint", type=str, default="")
    return ap.parse_args(argv)


def save_checkpoint(eng: Engine, path: str, state: dict):
    os.makedirs(path, exist_ok=True)
    eng.cfg.save_pretrained(path)
    from safetensors.torch import save_file
    save_file({k: v.detach().cpu().contiguous() for k, v in eng.params.items()}, os.path.join(path, "model.safetensors"))
    torch.save({"m": eng.flat_m.cpu() if eng.flat_m is not None else None,
                "v": eng.flat_v.cpu() if eng.flat_v is not None else None,
                "opt_step": eng.opt_step, "layout_version": eng.layout.version}, os.path.join(path, "optimizer.pt"))
    with open(os.path.join(path, "trainer_state.json"), "w") as f:
        json.dump(state, f)


def load_checkpoint(eng: Engine, path: str) -> dict:
    from safetensors.torch import load_file
    eng.load_state_dict(load_file(os.path.join(path, "model.safetensors")))
    opt = torch.load(os.path.join(path, "optimizer.pt"), map_location="cpu")
    if opt["m"] is not None:
        # (the moments are stored flat: a file written under another parameter order is re-ordered by name)
        eng.flat_m = eng.layout.adopt(opt["m"], eng.cfg, opt.get("layout_version")).to(eng.device)
        eng.flat_v = eng.layout.adopt(opt["v"], eng.cfg, opt.get("layout_version")).to(eng.device)
    eng.opt_step = int(opt["opt_step"])
    with open(os.path.join(path, "trainer_state.json")) as f:
        return json.load(f)


@torch.no_grad()
def evaluate_loss(eng: Engine, samples, coll, batch_size: int, rank: int, world: int) -> float:
    """eval_loss of the HF Trainer over the validation set (train_SMB_decoder.py:409-416, eval_strategy="epoch"):
    the mean over samples of their batch's mean loss; every rank takes every world-th batch."""
    from . import data as gdata
    tot = torch.zeros(2, dtype=torch.float64, device=eng.device)
    for idx in gdata.batches(len(samples), batch_size, rank=rank, world=world):
        b = coll.train(samples, idx)
        loss, _ = eng.forward(b["input_ids"], b["attention_mask"], b["actions"], labels=b["labels"], train=False,
                              session_ids=b.get("session_ids"), extended_session_ids=b.get("extended_session_ids"))
        tot[0] += loss.double() * len(idx)
        tot[1] += len(idx)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(tot)
    return float(tot[0] / tot[1].clamp_min(1))


class Prefetcher:
    """Collates the micro-batches of the next optimizer steps on a background thread, into PINNED host tensors, so
    that the launching thread only issues asynchronous host-to-device copies: with real data the reference's
    DataLoader workers do this job; collating on the launching thread cost ~85 ms in front of a 378 ms step."""

    def __init__(self, make_step, first: int, last: int, depth: int):
        import queue
        import threading
        self.q = queue.Queue(maxsize=max(1, depth))
        self.err = None

        def work():
            try:
                for step in range(first, last):
                    micro = make_step(step)
                    n_local = float(sum(int((b["labels"][:, 1:] != -100).sum()) for b in micro))
                    micro = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in b.items()} for b in micro]
                    self.q.put((step, micro, n_local))
            except BaseException as e:                    # surfaced by next()
                self.err = e
            self.q.put(None)
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def next(self, device):
        item = self.q.get()
        if item is None:
            raise self.err if self.err is not None else StopIteration
        step, micro, n_local = item
        dev = [{k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()} for b in micro]
        return step, dev, n_local


def main(argv=None):
    args = parse_args(argv)
    if args.fp16:
        raise SystemExit("--fp16 is not built (the MI355X path has fp32 and bf16; the reference's recipe uses neither "
                         "loss scaling nor fp16 kernels of its own)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", local_rank))
    variant = "session" if args.backbone == "Qwen3SessionMulti" else "multi"
    accum = args.gradient_accumulation_steps
    real = None
    if args.data_path:
        # load_SMB_datasets(..., tasks) for the explicit-decoder task family (loading_SMB.py:39-55)
        from . import data as gdata
        task = args.tasks.lower()
        ds = gdata.SMBData(args.data_path, args.dataset, args.index_file)
        only_response = False
        if task.startswith("smb_explicit_decoder"):
            augment = int(task.split("_")[3]) if task != "smb_explicit_decoder" else None
            samples = ds.train_samples(args.max_his_len, augment=augment)
        elif task == "smb_explicit":
            # one sample per interaction, loss on the target item only (loading_SMB.py:24-38, train_SMB_decoder.py:271)
            samples = ds.train_samples_per_interaction(args.max_his_len)
            only_response = True
        else:
            raise NotImplementedError(f"tasks={args.tasks}: smb_explicit and smb_explicit_decoder[_N] are built")
        coll = gdata.Collator(ds)
        cfg = gdata.model_config(ds, args.max_his_len)
        per_step = args.per_device_batch_size * accum * world
        args.steps_per_epoch = max(1, len(samples) // per_step)            # drop_last, as the window needs equal shards
        real = dict(samples=samples, coll=coll, valid=ds.valid_samples(args.max_his_len), only_response=only_response)
        if rank == 0:
            print(json.dumps({"dataset": args.dataset, "train_samples": len(samples), "vocab_size": cfg.vocab_size,
                              "steps_per_epoch": args.steps_per_epoch}), flush=True)
    else:
        cfg = synthetic_config(n_positions=args.max_his_len + 1)
    eng = Engine(cfg, device=f"cuda:{local_rank}", temperature=args.temperature, variant=variant,
                 dtype="bf16" if args.bf16 else "f32", matmul=None if args.bf16 else args.matmul)
    eng.init_weights(seed=args.seed)
    # one workspace for the longest batch the collator can produce: shorter batches bind views of it, nothing is
    # allocated inside the loop
    eng.reserve(args.per_device_batch_size, (args.max_his_len + 1) * cfg.num_positions, train=True)
    eng.base_seed = args.seed * 1000 + rank
    reducer = GradAllReducer(eng.flat_g, eng.layout, cfg.num_hidden_layers) if world > 1 else None
    total_steps = args.epochs * args.steps_per_epoch
    warmup = warmup_steps_for(total_steps, args.warmup_ratio)
    state = {"global_step": 0, "epoch": 0, "log_history": [], "best_metric": None, "best_model_checkpoint": None,
             "evals_without_improvement": 0}
    if args.resume_from_checkpoint:
        state = load_checkpoint(eng, args.resume_from_checkpoint)
    items = args.max_his_len + 1
    t_last, seq_since = time.time(), 0
    epoch_order = {}

    def real_micro(step):
        """micro-batches of this rank for one optimizer step: a per-epoch permutation (same on every rank), cut into
        windows of world * accum * batch samples, rank r taking the r-th slice of every micro-batch group"""
        import numpy as np
        ep, k = divmod(step, args.steps_per_epoch)
        if ep not in epoch_order:
            epoch_order.clear()
            epoch_order[ep] = np.random.RandomState(args.seed + ep).permutation(len(real["samples"]))
        bs = args.per_device_batch_size
        win = epoch_order[ep][k * bs * accum * world:(k + 1) * bs * accum * world]
        return [real["coll"].train(real["samples"], win[(a * world + rank) * bs:(a * world + rank + 1) * bs],
                                   only_train_response=real["only_response"]) for a in range(accum)]

    def make_step(step):
        if real is not None:
            return real_micro(step)
        return [synthetic.make_batch(args.per_device_batch_size, items, 256, 3, ragged=args.ragged,
                                     seed=args.seed + 7919 * (step * accum + a) + 104729 * rank,
                                     session_mean=4.0 if variant == "session" else None) for a in range(accum)]

    feed = Prefetcher(make_step, state["global_step"], total_steps, args.prefetch)
    while state["global_step"] < total_steps:
        step, micro, n_local = feed.next(eng.device)
        assert step == state["global_step"]
        # label tokens of the whole accumulation window, summed over all ranks ON THE DEVICE (HF num_items_in_batch with
        # average_tokens_across_devices; no host read in the steady state), gradients accumulated over the window and
        # reduced during the last micro-batch's backward: Engine.train_window
        lr = cosine_with_warmup(step, args.learning_rate, warmup, total_steps)
        losses = eng.train_window(micro, lr, n_items=n_local, reducer=reducer, weight_decay=args.weight_decay,
                                  max_norm=args.max_grad_norm)
        loss = losses[-1]
        loss_sum = float(sum(float(x) for x in losses)) if (step + 1) % args.logging_step == 0 else 0.0
        state["global_step"] = step + 1
        seq_since += args.per_device_batch_size * accum * world
        if (step + 1) % args.logging_step == 0 or step + 1 == total_steps:
            torch.cuda.synchronize()
            eng.check_inputs()
            now = time.time()
            rec = {"step": step + 1, "loss": loss_sum * world if loss_sum else float(loss) * world * accum,
                   "grad_norm": float(eng.grad_norm), "learning_rate": lr,
                   "train_samples_per_second": seq_since / max(now - t_last, 1e-9)}
            state["log_history"].append(rec)
            if rank == 0:
                print(json.dumps(rec), flush=True)
            t_last, seq_since = now, 0
        if (step + 1) % args.steps_per_epoch == 0:
            state["epoch"] = (step + 1) // args.steps_per_epoch
            stop = False
            ck = os.path.join(args.output_dir, f"checkpoint-{step + 1}") if args.output_dir else None
            if real is not None and len(real["valid"]) > 0:
                # evaluation + best-checkpoint bookkeeping + early stopping, as the reference's Trainer is set up
                # (eval/save per epoch, load_best_model_at_end, EarlyStoppingCallback(patience), save_total_limit)
                ev = evaluate_loss(eng, real["valid"], real["coll"], args.per_device_batch_size, rank, world)
                rec = {"step": step + 1, "epoch": state["epoch"], "eval_loss": ev}
                state["log_history"].append(rec)
                if rank == 0:
                    print(json.dumps(rec), flush=True)
                if state["best_metric"] is None or ev < state["best_metric"]:
                    state["best_metric"], state["best_model_checkpoint"] = ev, ck
                    state["evals_without_improvement"] = 0
                else:
                    state["evals_without_improvement"] += 1
                    stop = state["evals_without_improvement"] >= args.patience
            if ck and rank == 0:
                save_checkpoint(eng, ck, state)
                import shutil
                kept = sorted((d for d in os.listdir(args.output_dir) if d.startswith("checkpoint-")),
                              key=lambda d: int(d.split("-")[1]))
                best = os.path.basename(state["best_model_checkpoint"]) if state["best_model_checkpoint"] else None
                for d in kept[:-1]:
                    if len([x for x in os.listdir(args.output_dir) if x.startswith("checkpoint-")]) <= args.save_total_limit:
                        break
                    if d != best:
                        shutil.rmtree(os.path.join(args.output_dir, d), ignore_errors=True)
            if stop:
                break
    if real is not None and args.output_dir and state.get("best_model_checkpoint"):
        # load_best_model_at_end + trainer.save_model(output_dir) (train_SMB_decoder.py:417, 446)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        from safetensors.torch import load_file
        eng.load_state_dict(load_file(os.path.join(state["best_model_checkpoint"], "model.safetensors")))
        if rank == 0:
            eng.cfg.save_pretrained(args.output_dir)
            from safetensors.torch import save_file
            save_file({k: v.detach().cpu().contiguous() for k, v in eng.params.items()},
                      os.path.join(args.output_dir, "model.safetensors"))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return state


if __name__ == "__main__":
    main()
