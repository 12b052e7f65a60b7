"""GPU: the train / eval harness end to end on a tiny configuration (reference call stacks SURVEY.md section 3.1 / 3.2)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hdiff_amd  # noqa: E402
from hdiff_amd.MainCondition import main  # noqa: E402


def test_train_then_eval_tiny(tmp_path):
    cfg = {
        "state": "train", "epoch": 10, "batch_size": 4, "T": 6, "channel": 32, "channel_mult": [1, 2], "num_res_blocks": 1,
        "dropout": 0.1, "lr": 2e-4, "multiplier": 2.5, "beta_1": 1e-4, "beta_T": 0.028, "img_size": 16, "grad_clip": 1.,
        "device": "cuda:0", "w": 1.8, "save_dir": str(tmp_path / "ckpt"), "training_load_weight": None,
        "test_load_weight": "ckpt_9_.pt", "sampled_dir": str(tmp_path / "samples"),
        "sampledNoisyImgName": "noisy.png", "sampledImgName": "sampled.png", "nrow": 4,
        "dataset": "synthetic", "synthetic_size": 24, "num_labels": 3, "num_workers": 0, "max_steps_per_epoch": 3,   # epoch >= 10: warm_epoch = epoch // 10 must be > 0, as in the reference
    }
    np.random.seed(0)
    torch.manual_seed(0)
    history = main(cfg)
    assert len(history) == 10 * 3 and all(np.isfinite(history))
    assert np.mean(history[-4:]) < np.mean(history[:4])                 # the loss moves down
    sd = torch.load(os.path.join(cfg["save_dir"], "ckpt_9_.pt"), map_location="cpu")
    assert "downblocks.0.attn.in_proj_weight" in sd and sd["head.weight"].shape == (32, 3, 3, 3)
    cfg2 = dict(cfg, state="eval", batch_size=8)
    imgs = main(cfg2)
    assert tuple(imgs.shape) == (8, 3, 16, 16) and float(imgs.min()) >= 0 and float(imgs.max()) <= 1
    assert os.path.isfile(os.path.join(cfg["sampled_dir"], "sampled.png"))
    assert os.path.isfile(os.path.join(cfg["sampled_dir"], "noisy.png"))


def test_bench_json_contract_small():
    """bench.py prints ONE JSON line with the driver's fields (run at 64x64, batch 2 so that it takes seconds)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--size", "64", "--batch", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches_timed"] == 2 * 2 and r["avg_launch_ms"] > 0
    alt = d["config"]["other_contract_mode"]
    assert alt["contract"] == "bf16x3" and alt["ms_per_step"] > 0
