"""Learning-rate schedule of the reference run: HF ``get_cosine_schedule_with_warmup``
(``lr_scheduler_type='cosine'``, ``warmup_ratio`` from ref:SeqRec/tasks/train_SMB_decoder.py:396-428)."""
import math


def cosine_with_warmup(step: int, base_lr: float, warmup_steps: int, total_steps: int) -> float:
    """LR for the optimizer step taken after ``step`` completed steps (0-based, like LambdaLR)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    progress = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


def warmup_steps_for(total_steps: int, warmup_ratio: float) -> int:
    return math.ceil(total_steps * warmup_ratio)       # TrainingArguments.get_warmup_steps
