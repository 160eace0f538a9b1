"""Launch times of the fp32 cross-entropy kernels at T = B x 505 rows of V = 1041 logits (ld 1056): GB/s of the bytes they move.
python tools/time_ce.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gamer_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S, V, ld = 505, 1041, 1056
dev = "cuda"
logits = torch.randn(B * S, ld, device=dev)
labels = torch.randint(0, V, (B, S), device=dev)
lse, rl = torch.empty(B * S, device=dev), torch.empty(B * S, device=dev)
ls, cnt = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
tf = t(lambda: ops.ce_fwd(logits, ld, labels, V, 0.7, -100, lse, rl, ls, cnt))
tb = t(lambda: ops.ce_bwd(logits, ld, labels, V, 0.7, -100, lse, cnt, 0.0, 1.0))
by = B * S * V * 4 * 2
print(f"ce_fwd (+ reduce) {tf:.3f} ms = {by / tf / 1e6:.0f} GB/s   ce_bwd {tb:.3f} ms = {by / tb / 1e6:.0f} GB/s")
