"""One rank of tests/test_dp_gpu.py: the real HIP engine under data parallelism, two ranks sharing cuda:0 over gloo.

usage: python dp_gpu_worker.py RANK WORLD PORT ACCUM DTYPE OUT.json
Every rank builds the same weights, takes rows r, r + world, ... of every global micro-batch of ONE accumulation window
(ragged rows: the label counts of the shards differ), and runs Engine.train_window with a GradAllReducer - the code path
of gamer_amd/train.py and bench.py --gpus N.  Rank 0 then repeats the window alone on the concatenated batches with a
second engine and writes the differences (reduced flat gradient, post-step parameters, losses).
ref: DDP semantics of ref:SeqRec/tasks/multi_gpu.py:53, ref:SeqRec/tasks/train_SMB_decoder.py:420."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, accum = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dtype, out_path = sys.argv[5], sys.argv[6]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from gamer_amd import synthetic
    from gamer_amd.config import synthetic_config
    from gamer_amd.dp import GradAllReducer
    from gamer_amd.engine import Engine

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = synthetic_config(num_hidden_layers=4, behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3],
                               sparse_layers_decoder=[0, 1, 2, 3], hidden_size=128, num_attention_heads=2,
                               num_key_value_heads=1, head_dim=64, intermediate_size=256, moe_intermediate_size=128,
                               codebook=16)
        cfg.dropout_rate = 0.0
        cfg.attention_dropout = 0.0
        rows, items = 8, 9
        window = [synthetic.make_batch(rows, items, 16, 3, seed=70 + a, pad_rows={1: 3, 4: 5, 6: 1}) for a in range(accum)]
        mine = [{k: v[torch.arange(rank, rows, world)] for k, v in b.items()} for b in window]

        def engine():
            e = Engine(cfg, device="cuda:0", temperature=0.7, dtype=dtype)
            e.init_weights(seed=11)
            return e

        eng = engine()
        sd0 = {k: v.detach().cpu().clone() for k, v in eng.params.items()}      # (for the oracle: the weights of window 1)
        red = GradAllReducer(eng.flat_g, eng.layout, cfg.num_hidden_layers)
        assert red.world == world and red.stage          # gloo + device gradients: the host-staged bucket path
        ref = engine() if rank == 0 else None
        res = {"rank": rank, "loss": [], "ref_loss": [], "grad_rel": [], "param_rel": [], "grad_norm": [], "grad_norm_ref": [],
               "param_max_diff_vs_rank0": []}
        # two consecutive windows: the accumulation state (zero_grad, Adam moments, step count) carries over
        for step in range(2):
            losses = eng.train_window(mine, 5e-4, reducer=red)
            torch.cuda.synchronize()
            res["loss"] += [float(x) for x in losses]
            res["grad_norm"].append(float(eng.grad_norm))
            # all ranks must hold the same parameters after the step
            p = eng.flat_p.cpu()
            other = p.clone()
            dist.broadcast(other, src=0)
            res["param_max_diff_vs_rank0"].append(float((p - other).abs().max()))
            if rank == 0:
                # the same window alone on the concatenated batches
                ref_losses = ref.train_window(window, 5e-4)
                torch.cuda.synchronize()
                g, gr = eng.flat_g.cpu().double(), ref.flat_g.cpu().double()
                res["grad_rel"].append(float((g - gr).abs().max() / gr.abs().max()))
                res["grad_norm_ref"].append(float(ref.grad_norm))
                pr = ref.flat_p.cpu()
                dp_ = (p.double() - pr.double()).abs()
                res["param_rel"].append(float(dp_.max() / pr.double().abs().max()))
                # share of the parameters that moved apart by more than rounding (2e-6 of the largest parameter)
                res.setdefault("param_frac_off", []).append(float((dp_ > 2e-6 * pr.double().abs().max()).double().mean()))
                # the well-conditioned elements (clipped gradient far above AdamW's eps, where lr g / (|g| + eps) does not amplify a
                # summation-order difference): the bar of rounds 2-3 stays on them
                clip = min(1.0, 1.0 / max(float(ref.grad_norm), 1e-30))
                strong = (gr.abs() * clip) > 1e-5
                res.setdefault("param_rel_strong", []).append(float(dp_[strong].max() / pr.double().abs().max()) if bool(strong.any()) else 0.0)
                res.setdefault("strong_frac", []).append(float(strong.double().mean()))
                res["ref_loss"] += [float(x) for x in ref_losses]
                if step == 0:
                    # ... and tied DIRECTLY to the CPU oracle: the gradient two HIP ranks reduced over the process group against
                    # oracle.loss_and_grads on the whole (unsharded) window with the window's label count as the normaliser
                    from oracle import qwen3multi_oracle as orc
                    ocfg = orc.OracleConfig.from_dict(cfg.to_dict())
                    n_items = float(sum(int((b["labels"][:, 1:] != -100).sum()) for b in window))
                    tot = None
                    for b in window:
                        _, grads, _ = orc.loss_and_grads(sd0, ocfg, b, temperature=0.7, num_items_in_batch=n_items,
                                                         amp=(dtype == "bf16"))
                        tot = grads if tot is None else {k: tot[k] + g for k, g in grads.items()}
                    views = eng.layout.views(eng.flat_g)
                    worst, worst_norm = 0.0, 0.0
                    for k, g in tot.items():
                        got = views[k].cpu().double()
                        worst = max(worst, float((got - g.double()).abs().max() / g.double().abs().max().clamp_min(1e-30)))
                        worst_norm = max(worst_norm, abs(float(got.norm()) - float(g.double().norm())) / max(float(g.double().norm()), 1e-30))
                    res["grad_rel_oracle"], res["grad_norm_rel_oracle"] = worst, worst_norm
        # the sum over ranks of the per-rank losses (each = its CE sum / GLOBAL count) is the single-rank loss
        ls = torch.tensor(res["loss"], dtype=torch.float64)
        dist.all_reduce(ls)
        res["loss_sum_over_ranks"] = ls.tolist()
        with open(out_path, "w") as f:
            json.dump(res, f)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
