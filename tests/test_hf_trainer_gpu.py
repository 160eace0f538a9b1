"""The drop-in claim of INTEGRATION.md, exercised: the reference trains through HF ``Trainer``
(ref:SeqRec/tasks/train_SMB_decoder.py:396-444).  Two optimizer steps of the stock ``transformers.Trainer`` on
``gamer_amd.modeling.Qwen3MultiWithTemperature`` must leave the same weights as two ``Engine.train_step`` calls
(clip 1.0, adamw_torch, weight decay 0.01 except the norm weights, loss = sum CE / num_items_in_batch)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hf_trainer_drives_the_module(golden, tmp_path):
    transformers = pytest.importorskip("transformers")
    from torch.utils.data import SequentialSampler
    from gamer_amd import synthetic
    from gamer_amd.config import Qwen3MultiConfig
    from gamer_amd.engine import Engine
    from gamer_amd.modeling import Qwen3MultiWithTemperature
    from oracle import qwen3multi_oracle as orc

    z, meta = golden("small")
    cfgd = {**meta["config"], "dropout_rate": 0.0, "attention_dropout": 0.0}
    ocfg = orc.OracleConfig.from_dict(cfgd)
    sd = orc.init_state_dict(ocfg, seed=3)
    lr, bs = 5e-4, 3
    data = synthetic.make_batch(2 * bs, 9, meta["codebook"], 3, seed=123, pad_rows={1: 2, 4: 5})
    keys = ("input_ids", "attention_mask", "actions", "labels", "session_ids", "extended_session_ids")

    # ---- reference run of this repo: the fused engine step
    eng = Engine(Qwen3MultiConfig(**cfgd), temperature=0.7)
    eng.load_state_dict(sd)
    for s in range(2):
        eng.train_step({k: data[k][s * bs:(s + 1) * bs] for k in keys}, lr)
    want = {k: v.detach().cpu().clone() for k, v in eng.params.items()}

    # ---- the same two steps through transformers.Trainer
    model = Qwen3MultiWithTemperature(Qwen3MultiConfig(**cfgd))
    model.set_hyper(0.7)
    model.load_state_dict({**sd, "lm_head.weight": sd["model.embed_tokens.weight"]})
    names = [n for n, _ in model.named_parameters()]
    assert set(names) == set(sd), "named_parameters() must yield the reference's parameter names"

    class Rows(torch.utils.data.Dataset):
        def __len__(self):
            return 2 * bs

        def __getitem__(self, i):
            return {k: data[k][i] for k in keys}

    class OrderedTrainer(transformers.Trainer):
        def _get_train_sampler(self, *a, **k):
            return SequentialSampler(self.train_dataset)

    args = transformers.TrainingArguments(
        output_dir=str(tmp_path), per_device_train_batch_size=bs, max_steps=2, learning_rate=lr,
        lr_scheduler_type="constant", warmup_steps=0, optim="adamw_torch", weight_decay=0.01, max_grad_norm=1.0,
        adam_beta1=0.9, adam_beta2=0.999, adam_epsilon=1e-8, report_to=[], save_strategy="no", logging_steps=1,
        remove_unused_columns=False, dataloader_num_workers=0, seed=0, disable_tqdm=True)
    trainer = OrderedTrainer(model=model, args=args, train_dataset=Rows(),
                             data_collator=lambda rows: {k: torch.stack([r[k] for r in rows]) for k in keys})
    # the decay / no-decay split the optimizer will be built from
    decay = set(trainer.get_decay_parameter_names(model))
    assert all(("norm" in n) != (n in decay) for n in names), "norm weights (and only they) are excluded from weight decay"
    trainer.train()
    got = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    worst = 0.0
    for k, w in want.items():
        d = (got[k] - w).abs()
        moved = (w - sd[k]).abs()
        # same update: compare where the step is well conditioned (|dw| ~ lr), bound the rest by 2*lr
        assert float(d.max()) <= 2 * 2 * lr + 1e-7, k
        worst = max(worst, float(d.max() / (moved.max() + 1e-12)))
    assert worst < 5e-3, worst
    assert float(np.mean([float((got[k] - sd[k]).abs().max()) for k in want])) > 1e-5      # it did train
