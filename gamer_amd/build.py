"""Build libgamer_hip.so (gfx950) in-tree with hipcc.  `python -m gamer_amd.build [--force]`."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgamer_hip.so")
TORCH_LIB = os.path.join(LIBDIR, "libgamer_torch.so")          # TORCH_LIBRARY(gamer, ...) wrappers over the C ABI
SOURCES = ["prep.hip", "inject.hip", "elementwise.hip", "gemm.hip", "gemm_as.hip", "gemm_wg.hip", "gemm_os.hip", "gemm_bf16.hip", "attention.hip", "attention_split.hip", "attention_res.hip", "attention_bf16.hip", "optim.hip", "decode.hip",
           "modules.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++20", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/gamer_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libgamer_hip.sha256")
    dig = _digest()
    if (not force and os.path.exists(LIB) and os.path.exists(TORCH_LIB) and os.path.exists(stamp)
            and open(stamp).read().strip() == dig):
        return LIB
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    build_torch_ops(verbose)
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


def build_torch_ops(verbose: bool = True) -> str:
    """csrc/torch_ops.cpp -> lib/libgamer_torch.so: host C++ only (no device code), compiled with g++ against the
    installed torch headers and linked to libgamer_hip.so next to it (rpath $ORIGIN)."""
    import torch
    tdir = os.path.dirname(torch.__file__)
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-O2", "-shared", "-fPIC", "-std=c++17", os.path.join(CSRC, "torch_ops.cpp"),
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
           f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include", "-I/opt/rocm/include",
           f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip",
           f"-L{LIBDIR}", "-lgamer_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tdir}/lib", "-o", TORCH_LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed for torch_ops.cpp:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr, file=sys.stderr)
    return TORCH_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
