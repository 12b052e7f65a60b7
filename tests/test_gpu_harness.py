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


def _json_line(res):
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    import json
    return json.loads(lines[0])


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract_small():
    """bench.py prints ONE JSON line with the driver's fields (run at 64x64, batch 2 so that it takes seconds), including
    the legs outside `value`: cpu_baseline + parity (the 128x128 forward on both sides) and the other BASELINE configs
    (toy-sized here: --extras-scale small runs the same code)."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--size", "64", "--batch", "2", "--extras-scale", "small"], capture_output=True, text=True, timeout=900)
    d = _json_line(res)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "configs"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"].startswith("f32") and d["data"] == "synthetic"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches_timed"] == 2 * 2 and r["avg_launch_ms"] > 0
    assert "kernels" in r["traffic_stamp"]
    # the line says what bounds the kernel -- the algorithmic fraction of the raw 16-bit peak, the issue model made of the
    # PMC instruction counts in profiles/ and the matrix-only floor at the power limit (None here: the counts are of the 256x256 shape), the box range
    assert abs(r["frac_vs_16bit_peak"] - r["achieved"] / 2516.6) < 1e-3
    for key in ("issue_model_ms", "frac_vs_issue_model", "matrix_only_floor_ms_at_power_limit", "frac_vs_matrix_only_floor"):
        assert key in r, key
    assert "denoising_steps_per_s" in d["box_range"] and "device_clock" in d
    assert "2xfp16" in d["dtype"] and "three at d_head 32" in d["dtype"]                  # the shipped d_head 32 kernel: fp16 pairs, three terms
    assert d["config"]["attention_contract"] == "bf16x3" and "3xbf16" in d["dtype"]        # the library's default mode
    alt = d["config"]["other_contract_mode"]
    assert alt["contract"] == "f32" and alt["ms_per_step"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["same_config"]["cpu_steps_per_s"] > 0
    par = d["parity"]
    assert par["max_abs"] <= par["tolerance_max_abs"], par
    assert par["modes"]["max_abs"] <= 1e-3 * max(1.0, par["modes"]["state_abs_max"]), par["modes"]     # the two contraction modes agree at the benchmark's own size
    assert par["end_to_end_C1"]["psnr_db"] >= 40.0
    cfgs = d["configs"]
    assert "error" not in cfgs, cfgs
    assert cfgs["C1"]["gpu_steps_per_s"] > 0 and cfgs["C2"]["steps_per_s"] > 0 and cfgs["C5"]["steps_per_s"] > 0
    assert cfgs["C3"]["all_grads_present"] and np.isfinite(cfgs["C3"]["loss"]) and cfgs["C3"]["s_per_step"] > 0


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher: the parent starts two fresh rank processes, relays rank 0's single line
    and the line says n_gpus 2 (both ranks share the box's one GPU over gloo: --rehearse-one-gpu; on a real node the same
    code runs one rank per GPU over RCCL)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--size", "64",
                          "--batch", "1", "--steps", "2", "--warmup", "1", "--no-kernel-pass"],
                         capture_output=True, text=True, timeout=900, env=env)
    d = _json_line(res)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]          # whole-job aggregate over both ranks
    assert "cpu_baseline" not in d and "configs" not in d                              # N = 1 legs only


def test_bench_train_gpus_2_starts_its_own_ranks():
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_train.py"), "--gpus", "2", "--rehearse-one-gpu",
                          "--size", "64", "--batch", "1", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, env=env)
    d = _json_line(res)
    assert d["n_gpus"] == 2 and d["gradient_exchange_bytes_per_rank"] >= 47_696_515 * 4 and np.isfinite(d["loss"])


def test_reference_entry_file_shape_runs_through_the_dropin(tmp_path):
    """A file with the reference's OWN import lines (MainCondition.py:1, TrainCondition.py:15-17) and its main(cfg) dispatch,
    run as a script after hdiff_amd.install_dropin(): trains a tiny model on the GPU, then samples from the checkpoint."""
    import subprocess
    import sys
    cfg = {
        "state": "train", "epoch": 10, "batch_size": 4, "T": 6, "channel": 32, "channel_mult": [1, 2], "num_res_blocks": 1,
        "dropout": 0.1, "lr": 2e-4, "multiplier": 2.5, "beta_1": 1e-4, "beta_T": 0.028, "img_size": 16, "grad_clip": 1.,
        "device": "cuda:0", "w": 1.8, "save_dir": str(tmp_path / "ckpt"), "training_load_weight": None,
        "test_load_weight": "ckpt_9_.pt", "sampled_dir": str(tmp_path / "samples"),
        "sampledNoisyImgName": "noisy.png", "sampledImgName": "sampled.png", "nrow": 4,
        "dataset": "synthetic", "synthetic_size": 8, "num_labels": 3, "num_workers": 0, "max_steps_per_epoch": 1,
    }
    script = tmp_path / "RefShapedMain.py"
    script.write_text(
        "from DiffusionFreeGuidence.TrainCondition import train, eval\n"
        "from DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler, GaussianDiffusionTrainer\n"
        "from DiffusionFreeGuidence.ModelCondition import UNet\n"
        "from Scheduler import GradualWarmupScheduler\n"
        "import json, sys\n"
        "def main(model_config=None):\n"
        "    modelConfig = model_config\n"
        "    if modelConfig['state'] == 'train':\n"
        "        train(modelConfig)\n"
        "    else:\n"
        "        eval(modelConfig)\n"
        "cfg = json.load(open(sys.argv[1]))\n"
        "main(cfg)\n"
        "main(dict(cfg, state='eval', batch_size=8))\n"
        "assert UNet.__module__.startswith('hdiff_amd.')\n"
        "print('DROPIN_OK')\n")
    import json
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    boot = (f"import sys, runpy; sys.path.insert(0, {ROOT!r}); import hdiff_amd; hdiff_amd.install_dropin(); "
            f"sys.argv = [{str(script)!r}, {str(tmp_path / 'cfg.json')!r}]; runpy.run_path({str(script)!r}, run_name='__main__')")
    res = subprocess.run([sys.executable, "-c", boot], capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert res.returncode == 0 and "DROPIN_OK" in res.stdout, res.stderr[-3000:]
    assert os.path.isfile(os.path.join(cfg["save_dir"], "ckpt_9_.pt"))
    assert os.path.isfile(os.path.join(cfg["sampled_dir"], "sampled.png"))
