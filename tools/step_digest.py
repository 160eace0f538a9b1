"""GPU: loss and a digest of the flat gradient after one training step (forward + backward, dropout on) at B x 505 tokens - two builds of
the library that claim the same bits print the same line: GAMER_LIB_PATH=<variant.so> python tools/step_digest.py [B] [f32|bf16] [matmul]"""
import hashlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import synthetic
from gamer_amd.config import synthetic_config
from gamer_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
mm = sys.argv[3] if len(sys.argv) > 3 else "split3"
cfg = synthetic_config()
eng = Engine(cfg, temperature=0.7, dtype=dt, matmul=mm)
eng.init_weights(seed=0)
batch = {k: v.cuda() for k, v in synthetic.make_batch(B, 101, 256, 3, seed=5, behavior_probs=[0.7, 0.25, 0.05]).items()}
for it in range(2):
    loss, logits = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"], train=True)
    eng.backward()
torch.cuda.synchronize()
g = eng.flat_g.detach().cpu().contiguous()
print(f"B={B} {dt} {mm}: loss {float(loss):.9f} logits {hashlib.sha256(logits.detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]} "
      f"grad {hashlib.sha256(g.numpy().tobytes()).hexdigest()[:16]} |g| {float(g.double().norm()):.9f}", flush=True)
