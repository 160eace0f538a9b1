"""The training harness end to end on the GPU: loss goes down on a fixed synthetic stream, checkpoints
round-trip, and the RCCL code path runs (single rank)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_harness_learns_and_resumes(tmp_path):
    from gamer_amd import train
    args = ["--max_his_len", "20", "--per_device_batch_size", "16", "--gradient_accumulation_steps", "2", "--epochs", "2",
            "--steps_per_epoch", "12", "--logging_step", "4", "--warmup_ratio", "0.1", "--output_dir", str(tmp_path)]
    state = train.main(args)
    losses = [r["loss"] for r in state["log_history"] if "loss" in r]
    assert losses[-1] < losses[0] - 0.3, losses            # 24 steps at lr 5e-4 move the loss well below ln(V)
    assert all(r["grad_norm"] > 0 for r in state["log_history"])
    ck = os.path.join(str(tmp_path), "checkpoint-12")
    assert os.path.exists(os.path.join(ck, "model.safetensors")) and os.path.exists(os.path.join(ck, "config.json"))
    state2 = train.main(args + ["--resume_from_checkpoint", ck])
    assert state2["global_step"] == 24
    # same data stream + restored weights/moments/step counter -> same trajectory up to dropout RNG
    assert abs(state2["log_history"][-1]["loss"] - losses[-1]) < 0.15


def test_train_harness_bf16_flag_and_fp16_refusal(tmp_path):
    """--bf16 (the reference's AMP flag, train_SMB_decoder.py:114-118) trains through the bf16 kernels with fp32 master
    weights: the loss falls like the fp32 run's on the same stream; --fp16 is refused, not ignored."""
    from gamer_amd import train
    base = ["--max_his_len", "20", "--per_device_batch_size", "16", "--gradient_accumulation_steps", "2", "--epochs", "1",
            "--steps_per_epoch", "24", "--logging_step", "4", "--warmup_ratio", "0.1"]
    s16 = train.main(base + ["--bf16"])
    s32 = train.main(base)
    l16 = [r["loss"] for r in s16["log_history"] if "loss" in r]
    l32 = [r["loss"] for r in s32["log_history"] if "loss" in r]
    assert l16[-1] < l16[0] - 0.3, l16
    assert abs(l16[-1] - l32[-1]) < 0.15, (l16, l32)
    assert all(r["train_samples_per_second"] > 0 for r in s16["log_history"] if "loss" in r)
    with pytest.raises(SystemExit):
        train.main(base + ["--fp16"])


@pytest.mark.parametrize("backbone", ["Qwen3Multi", "Qwen3SessionMulti"])
def test_train_harness_on_a_dataset_in_the_reference_format(tmp_path, backbone):
    """--data_path/--dataset/--tasks as upstream: dataset directory (written in the reference's on-disk format) ->
    gamer_amd.data samples -> collated id tensors -> engine; the loss falls and a checkpoint with the dataset's
    vocabulary is written."""
    from gamer_amd import synthetic, train
    synthetic.write_smb_dataset(str(tmp_path / "data"), "Syn", n_users=300, n_items=200, codebook=32, max_sessions=9, seed=1)
    out = tmp_path / "ckpt"
    state = train.main(["--data_path", str(tmp_path / "data"), "--dataset", "Syn", "--tasks", "smb_explicit_decoder_2",
                        "--backbone", backbone, "--max_his_len", "20", "--per_device_batch_size", "32",
                        "--gradient_accumulation_steps", "1", "--epochs", "6", "--logging_step", "5",
                        "--output_dir", str(out)])
    losses = [r["loss"] for r in state["log_history"] if "loss" in r]
    assert state["global_step"] >= 30 and losses[-1] < losses[0] - 0.3, losses
    # per-epoch validation loss, best checkpoint, at most save_total_limit checkpoints, best model in output_dir
    evals = [r["eval_loss"] for r in state["log_history"] if "eval_loss" in r]
    assert len(evals) == 6 and evals[-1] < evals[0] and state["best_metric"] == min(evals)
    cks = sorted(d for d in os.listdir(out) if d.startswith("checkpoint-"))
    assert 1 <= len(cks) <= 3 and os.path.basename(state["best_model_checkpoint"]) in cks
    cfg = json.load(open(os.path.join(out, "config.json")))
    assert cfg["vocab_size"] == 14 + 4 * 32 + 3 and len(cfg["behavior_maps"]) == 3
    # the evaluation task on the same directory: test split -> prompts -> trie-constrained beams -> metrics
    from gamer_amd import data as gdata, modeling
    from gamer_amd.evaluate import evaluate_dataset
    model = getattr(modeling, backbone + "WithTemperature").from_pretrained(str(out))
    ds = gdata.SMBData(str(tmp_path / "data"), "Syn")
    res = evaluate_dataset(model.engine, ds, 20, num_beams=10, batch_size=50,
                           metric_list=("hit@1", "hit@10", "ndcg@10", "recall@10"))
    test = ds.test_samples(20)
    for beh in ds.behaviors:
        assert res[beh]["samples"] == len(test.filter_by_behavior(beh))
        assert 0.0 <= res[beh]["hit@1"] <= res[beh]["hit@10"] <= 1.0 and 0.0 <= res[beh]["ndcg@10"] <= 1.0


def test_bench_rccl_path_single_rank():
    env = dict(os.environ, GAMER_BENCH_FORCE_DIST="1")
    env.pop("NCCL_DEBUG_FILE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "8", "--items", "21", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.strip().splitlines()[-1] == lines[0], "the JSON line must be the last line on stdout"
    line = json.loads(lines[0])
    assert line["unit"] == "sequences/s" and line["value"] > 0 and "roofline" in line, lines[0][:600]
    # the multi-GPU diagnostics ride on the same path: per-bucket all-reduce times, exposed communication, RCCL's log summary
    comm = line["communication"]
    assert len(comm["per_bucket"]) == 8 + 2 and all(b["ms"] > 0 and b["bytes"] > 0 for b in comm["per_bucket"]), comm
    assert comm["step_ms_with_reducer"] > 0 and comm["step_ms_without_collectives"] > 0 and comm["exposed_communication_ms"] >= 0, comm
    assert sum(b["bytes"] for b in comm["per_bucket"]) == 4 * 24535040 or line["config"]["seq_len"] != 505
    assert "rccl" in line and line["rccl"].get("lines", 0) > 0, line.get("rccl")
