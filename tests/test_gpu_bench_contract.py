"""-m gpu: the driver's contract with bench.py -- ONE JSON line with the agreed keys, a `roofline` object priced on algorithmic
bytes with the step-level fraction beside it, a `cpu_baseline` object from the oracle, and a value that follows from the timing."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_contract_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "8", "--imgsz", "320", "--lr", "0.0005"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "images/s" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "bytes_per_launch", "bytes_definition", "step", "dominant_launch",
              "top_conv", "ranking"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the headline is SURVEY.md 8(d)'s: algorithmic bytes of the dominant ALGORITHMIC kernel (a convolution / weight gradient / stem);
    # the largest group among all launches -- a BatchNorm pass, 0 algorithmic bytes -- sits under dominant_launch with its own bytes
    assert r["kernel"].startswith(("conv_", "stem_")) and r["bytes_per_launch"] > 0 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    dl = r["dominant_launch"]
    assert r["ranking"][0]["kernel"] == dl["kernel"] and abs(dl["own_frac"] - dl["own_achieved"] / 8000.0) < 1e-9
    assert r["top_conv"]["kernel"].startswith("conv_")
    assert 0 < r["step"]["frac"] < 1 and r["step"]["algorithmic_bytes_per_image"] == 367.1e6
    assert d["config"]["secondary"] is None  # only the full-size default run times configs[3] / configs[4] (test below)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "images/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert d["config"]["skipped_steps"] <= 8 and d["config"]["skipped_in_timed_steps"] == 0  # the loss-scale search ends before the timed steps
    assert d["config"]["capture_retries"] == 0  # one process on the GPU: the step graph verifies on its first capture
    assert d["config"]["loss_mode"] == "wiou+nwd" and d["config"]["ciou_images_per_s"] > 0 and c["bs16"] > 0


def test_bench_secondary_workloads_have_their_keys():
    """config.secondary of the default run: BASELINE.json configs[3] (LD train) and configs[4] (yolov8n-p2 1280x1280 batch 32 fused
    inference) timed by the functions the default run calls, with their 8(d) step fractions (run here directly: the full default
    bench takes a minute)."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    dev = torch.device("cuda", 0)
    p2 = b.secondary_p2_1280(dev, warmup=3, iters=5)
    for k in ("workload", "fps", "ms_per_forward", "latency_ms_per_image", "algorithmic_bytes_per_image", "step_frac"):
        assert k in p2, k
    assert abs(p2["fps"] - 32e3 / p2["ms_per_forward"]) < 1e-6 * p2["fps"] and 0 < p2["step_frac"] < 1 and "configs[4]" in p2["workload"]
    assert abs(p2["step_frac"] - p2["fps"] * 528.0e6 / 8e12) < 1e-9
    ld = b.secondary_ld(dev, steps=3)
    for k in ("workload", "images_per_s", "ms_per_step", "steps", "algorithmic_bytes_per_image", "step_frac"):
        assert k in ld, k
    assert abs(ld["images_per_s"] - 64e3 / ld["ms_per_step"]) < 1e-6 * ld["images_per_s"] and 0 < ld["step_frac"] < 1 and "configs[3]" in ld["workload"]


def test_bench_gpus_2_from_a_plain_python_start():
    """`python bench.py --gpus 2` with no launcher around it (what the driver runs on a multi-GPU node): bench.py starts
    torch.distributed.run itself as a child process before touching the GPU and relays rank 0's line (reference
    engine/trainer.py:607-627).  Rehearsed here with both ranks on the one GPU over gloo."""
    env = dict(os.environ, DY_REHEARSE_ON_ONE_GPU="1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
                        "--imgsz", "320", "--lr", "0.0005", "--no-cpu", "--probe", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-8000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["config"]["skipped_in_timed_steps"] == 0
