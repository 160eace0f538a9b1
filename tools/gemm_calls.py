"""GPU: every fp32 GEMM launch of one training step (Engine forward + backward at B x 505 tokens) with its shape, layout, epilogue and
HIP-event time, largest first, and which of the kernels (tile / gemm_as / gemm_os / gemm_wg) took it.  python tools/gemm_calls.py [B] [f32|bf16]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamer_amd import ops, synthetic
from gamer_amd.config import synthetic_config
from gamer_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = synthetic_config()
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
eng = Engine(cfg, temperature=0.7, dtype=DT)
eng.init_weights(seed=0)
batch = {k: v.cuda() for k, v in synthetic.make_batch(B, 101, 256, 3, seed=5, behavior_probs=[0.7, 0.25, 0.05]).items()}
rec = []
orig = ops.gemm
import ctypes
from gamer_amd import _lib
_fns = []
for _n in ("gamer_debug_gemm_as_launches", "gamer_debug_gemm_os_launches", "gamer_debug_gemm_wg_launches"):
    _f = getattr(_lib.load(), _n)
    _f.restype = ctypes.c_longlong
    _fns.append(_f)


def counters():
    return tuple(int(f()) for f in _fns)


def hooked(A, a_rs, a_ks, Bm, b_rs, b_ks, Cm, ldc, M, N, K, *a, **kw):
    names = ["alpha", "accumulate", "groups", "group_mode", "group_offsets", "strideB", "strideC", "kchunk", "resid", "row_map", "p_drop",
             "seed", "rowdot", "qknorm", "c_amax", "swiglu_bwd", "group_div", "sw_tbl", "swiglu_fwd"]
    k2 = dict(zip(names, a)); k2.update(kw)
    c0 = counters()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig(A, a_rs, a_ks, Bm, b_rs, b_ks, Cm, ldc, M, N, K, *a, **kw); e.record()
    c1 = counters()
    kern = "as" if c1[0] > c0[0] else "os" if c1[1] > c0[1] else "wg" if c1[2] > c0[2] else "tile"
    epi = "+".join(n for n in ("resid", "row_map", "rowdot", "qknorm", "swiglu_bwd", "swiglu_fwd", "sw_tbl") if k2.get(n) is not None) or "-"
    rec.append((s, e, M, N, K, a_ks, b_ks, k2.get("groups", 1), k2.get("group_mode", 0), bool(k2.get("accumulate", False)), epi, kern,
                str(A.dtype).replace("torch.", "")))
    return r


def step():
    loss, _ = eng.forward(batch["input_ids"], batch["attention_mask"], batch["actions"], labels=batch["labels"], train=True)
    eng.backward()


for it in range(3):
    if it == 2:
        ops.gemm = hooked
        rec.clear()
    step()
torch.cuda.synchronize()
ops.gemm = orig
rows = {}
for s, e, *key in rec:
    k = tuple(key)
    t = s.elapsed_time(e)
    n, tot = rows.get(k, (0, 0.0))
    rows[k] = (n + 1, tot + t)
print(f"B = {B}: {len(rec)} GEMM calls, {sum(v[1] for v in rows.values()):.1f} ms")
print("    M        N     K  a_ks b_ks groups mode accum epilogue                 kernel  calls  ms/call  ms/step  TFLOP/s")
for k, (n, tot) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    M, N, K, a_ks, b_ks, groups, mode, acc, epi, kern, dt = k
    fl = 2.0 * M * N * K
    print(f"{M:9d} {N:6d} {K:7d} {a_ks:4d} {b_ks:4d} {groups:5d} {mode:4d} {int(acc):5d} {epi:24s} {kern:6s} {n:5d} {tot / n:8.3f} {tot:8.2f} {fl / (tot / n) / 1e9:8.1f}")
