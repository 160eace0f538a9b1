"""bench.py --gpus N started without a launcher spawns N ranks itself (ref:scripts/train_SMB_decoder.sh:123-153)
and relays rank 0's line; driven here with a fake launcher, no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "helpers", "fake_launcher.py")


def _run(extra_env, *args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          env=env, timeout=300)


def test_spawn_relays_rank0_line():
    r = _run({"GAMER_BENCH_LAUNCHER": f"{sys.executable} {FAKE}"}, "--gpus", "4", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                          # ONE JSON line on stdout, the child's noise went to stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 4
    assert rec["argv"] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_spawn_fails_when_fewer_ranks_joined():
    r = _run({"GAMER_BENCH_LAUNCHER": f"{sys.executable} {FAKE} --lie 1"}, "--gpus", "2")
    assert r.returncode == 4
    assert r.stdout.strip() == ""


def test_gpus_mismatch_inside_a_launcher_is_an_error():
    r = _run({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "4")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_no_devices_is_an_error_not_a_one_rank_run():
    # this container has no GPU: asking for 2 must fail loudly instead of printing an n_gpus=1 record
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    r = _run({}, "--gpus", "2")
    assert r.returncode == 2 and r.stdout.strip() == ""


def test_batch_arithmetic():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--gpus", "8"])
    assert a.batch == 128 and not a.weak            # north star: global 1024 over 8 GPUs
    a = bench.parse_args(["--gpus", "8", "--weak"])
    assert a.batch == 1024 and a.weak               # BASELINE configs[2]
    a = bench.parse_args(["--batch", "128"])
    assert a.batch == 128 and a.weak
    with pytest.raises(SystemExit):
        bench.parse_args(["--gpus", "3"])           # 1024 is not divisible by 3
    cmd = bench.launcher_command(8, 1234, ["--gpus", "8"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
