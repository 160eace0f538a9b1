"""Per-shape time of the forward Linear GEMMs the activation-stationary kernel takes (q|k|v N=768, gate N=256, head N=1041; K=256) at
T = B x 505 tokens, through Engine.forward's own call sites: HIP-event time of the gemm_fwd family per forward, GAMER_GEMM_AS=0/1.
python tools/dev_gemm_as_time.py [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config
from gamer_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = synthetic_config()
eng = Engine(cfg, temperature=0.7)
eng.init_weights(seed=0)
batch = {k: v.cuda() for k, v in synthetic.make_batch(B, 101, 256, 3, seed=5, behavior_probs=[0.7, 0.25, 0.05]).items()}
rec = []
orig = ops.call
def timed(name, *a):
    if name != "gamer_gemm_f32_split": return orig(name, *a)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig(name, *a); e.record(); rec.append((s, e)); return r
for form in ("0", "1"):
    os.environ["GAMER_GEMM_AS"] = form
    ops.reload_env()          # (the library caches its switches)
    for it in range(3):
        if it == 2: ops.call = timed; rec.clear()
        eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"], train=True)
    torch.cuda.synchronize(); ops.call = orig
    ts = [s.elapsed_time(e) for s, e in rec]
    print(f"GAMER_GEMM_AS={form}: {len(ts)} GEMM launches, {sum(ts):.2f} ms: " + " ".join(f"{t:.3f}" for t in ts), flush=True)
