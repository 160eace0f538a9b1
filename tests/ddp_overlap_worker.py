"""One rank of tests/test_dp_gpu.py::test_module_under_ddp_with_the_engines_own_overlap: the nn.Module path wrapped in torch
DistributedDataParallel (as HF Trainer wraps it), two ranks sharing cuda:0 over gloo, gradients reduced by the engine's own
per-layer buckets during backward (model.enable_dp_overlap()) with DDP's reduction replaced by already_reduced_hook.
usage: python ddp_overlap_worker.py RANK WORLD PORT OUT.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from gamer_amd import synthetic
    from gamer_amd.config import synthetic_config
    from gamer_amd.dp import already_reduced_hook
    from gamer_amd.modeling import Qwen3MultiWithTemperature

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = synthetic_config(num_hidden_layers=4, behavior_injection_decoder=[0, 1], cross_attention_decoder=[2, 3],
                               sparse_layers_decoder=[0, 1, 2, 3], hidden_size=128, num_attention_heads=2,
                               num_key_value_heads=1, head_dim=64, intermediate_size=256, moe_intermediate_size=128, codebook=16)
        cfg.dropout_rate = 0.0
        cfg.attention_dropout = 0.0
        rows = 8
        batch = synthetic.make_batch(rows, 9, 16, 3, seed=91, pad_rows={1: 3, 6: 2})
        mine = {k: v[torch.arange(rank, rows, world)].cuda() for k, v in batch.items()}

        def model_():
            m = Qwen3MultiWithTemperature(cfg)
            m.set_hyper(0.7)
            m.train()
            return m

        model = model_().enable_dp_overlap()
        ddp = DDP(model, device_ids=[0], init_sync=False)
        ddp.register_comm_hook(None, already_reduced_hook)
        out = ddp(input_ids=mine["input_ids"], attention_mask=mine["attention_mask"], actions=mine["actions"], labels=mine["labels"])
        out.loss.backward()
        torch.cuda.synchronize()
        got = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
        res = {"rank": rank}
        # reference: the mean over ranks of each rank's own gradient (what DDP computes), from un-wrapped modules
        ref = model_()
        o = ref(input_ids=mine["input_ids"], attention_mask=mine["attention_mask"], actions=mine["actions"], labels=mine["labels"])
        o.loss.backward()
        torch.cuda.synchronize()
        worst = 0.0
        for k, p in ref.named_parameters():
            g = p.grad.detach().cpu().double()
            dist.all_reduce(g)
            g /= world
            worst = max(worst, float((got[k] - g).abs().max() / g.abs().max().clamp_min(1e-30)))
        res["worst_rel"] = worst
        res["n_params"] = len(got)
        with open(out_path, "w") as f:
            json.dump(res, f)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
