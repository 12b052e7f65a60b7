"""CPU-only checks: the C-ABI library loads and exports everything include/hdiff.h declares, the host-side classes keep
the reference's state_dict / buffers, and the product path refuses to run without an MI355X (no fallback)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import hdiff_amd
from hdiff_amd import _capi, engine as E
from hdiff_amd.DiffusionFreeGuidence import DiffusionCondition as DC
from hdiff_amd.DiffusionFreeGuidence import ModelCondition as MC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hdiff.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hdiff_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hdiff_amd.lib()
    assert lib.hdiff_abi_version() == 6
    syms = header_symbols()
    assert len(syms) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (hdiff_[a-z0-9_]+)", out))
    assert set(syms) <= exported, sorted(set(syms) - exported)
    assert set(_capi.EXPORTED_SYMBOLS) == set(syms), set(_capi.EXPORTED_SYMBOLS) ^ set(syms)
    assert lib.hdiff_device_count() >= 0


def test_argument_validation_without_gpu():
    """Entry points validate shapes on the host before any launch (no compute call is made here)."""
    lib = hdiff_amd.lib()
    assert lib.hdiff_mha_flash_fwd(1, 1, None, 1, 40, 8, 16, None) == -1      # head dim 5 unsupported
    assert b"head dim" in lib.hdiff_last_error()
    assert lib.hdiff_gn_stats(1, None, 48, 0, 1, 16, 32, 1, 1, None) == -1      # 48 channels / 32 groups
    d = _capi.ConvDesc()
    assert lib.hdiff_conv2d_fwd(d, None) == -1


def test_contraction_mode_switch():
    """hdiff_set_contraction_mode is process-wide, validated, and readable back (no launch is made here)."""
    lib = hdiff_amd.lib()
    start = lib.hdiff_get_contraction_mode()
    assert start in (0, 1)
    try:
        assert lib.hdiff_set_contraction_mode(1) == 0 and lib.hdiff_get_contraction_mode() == 1
        assert lib.hdiff_set_contraction_mode(7) == -1 and b"unknown mode" in lib.hdiff_last_error()
        assert lib.hdiff_get_contraction_mode() == 1
        hdiff_amd.set_contraction_mode("f32")
        assert hdiff_amd.get_contraction_mode() == "f32"
        with pytest.raises(ValueError):
            hdiff_amd.set_contraction_mode("fp8")
    finally:
        lib.hdiff_set_contraction_mode(start)


def test_product_path_refuses_cpu_tensors():
    m = MC.UNet(T=8, num_labels=3, ch=32, ch_mult=[1, 2], num_res_blocks=1, dropout=0.0).eval()
    x, t, lab = torch.zeros(1, 3, 16, 16), torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, t, lab)
    samp = DC.GaussianDiffusionSampler(m, 1e-4, 0.028, 8, w=1.8)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        samp(x, lab)
    tr = DC.GaussianDiffusionTrainer(m, 1e-4, 0.028, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr(x, lab)
    with pytest.raises(RuntimeError):
        E.Plan("cpu")


def test_state_dict_layout_and_seeded_init_match_reference():
    with open(os.path.join(GOLDEN, "state_dict_default.json")) as fh:
        ref = json.load(fh)
    d = np.load(os.path.join(GOLDEN, "unet_default64.npz"))
    cfg = json.loads(bytes(d["cfg_json"]).decode())
    torch.manual_seed(int(d["seed"][0]))
    m = MC.UNet(**cfg)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["entries"]
    assert sum(p.numel() for p in m.parameters()) == ref["n_params"]
    sd = m.state_dict()
    names = sorted(sd.keys())
    assert names == list(d["weight_names"])
    table = "time_embedding.timembedding.0.weight"
    for n, want in zip(names, d["weight_checksums"]):
        bits = sd[n].float().contiguous().reshape(-1).view(torch.int32).to(torch.int64)
        got = [int(bits.sum().item()), bits.numel(), int(bits[0].item()), int(bits[-1].item())]
        if n != table:
            assert got == list(want), n
    assert (sd[table][417] - torch.from_numpy(d["temb_row_417"])).abs().max().item() < 1e-4
    assert torch.all(sd["cond_embedding.condEmbedding.0.weight"][0] == 0)       # padding_idx row


def test_small_state_dict_loads_strict_and_buffers_bit_exact():
    u = np.load(os.path.join(GOLDEN, "unet_small.npz"))
    cfg = json.loads(bytes(u["cfg_json"]).decode())
    m = MC.UNet(**cfg)
    sd = {k[3:]: torch.from_numpy(u[k]) for k in u.files if k.startswith("sd/")}
    m.load_state_dict(sd, strict=True)
    s = np.load(os.path.join(GOLDEN, "schedules.npz"))
    for i in range(3):
        b1, bT, Tn = s[f"cfg{i}"]
        tr = DC.GaussianDiffusionTrainer(m, float(b1), float(bT), int(Tn))
        sa = DC.GaussianDiffusionSampler(m, float(b1), float(bT), int(Tn), w=1.8)
        assert [n for n, _ in tr.named_buffers(recurse=False)] == ["betas", "sqrt_alphas_bar", "sqrt_one_minus_alphas_bar"]
        assert [n for n, _ in sa.named_buffers(recurse=False)] == ["betas", "coeff1", "coeff2", "posterior_var"]
        for n in ("betas", "sqrt_alphas_bar", "sqrt_one_minus_alphas_bar"):
            assert getattr(tr, n).dtype == torch.float64
            assert np.array_equal(getattr(tr, n).numpy(), s[f"cfg{i}/trainer/{n}"])
        for n in ("betas", "coeff1", "coeff2", "posterior_var"):
            assert np.array_equal(getattr(sa, n).numpy(), s[f"cfg{i}/sampler/{n}"])
        ex = DC.extract(sa.coeff2, torch.from_numpy(s[f"cfg{i}/extract_t"]), (4, 3, 8, 8))
        assert np.array_equal(ex.numpy(), s[f"cfg{i}/extract_coeff2"])
        assert sa.model is m and sa.T == int(Tn) and sa.w == 1.8


def test_transposed_conv_phases_cover_every_tap_once():
    seen = set()
    for py in (0, 1):
        for px in (0, 1):
            t = E.tconv_phase_taps(py, px)
            assert len(t.dy) == (3 if py == 0 else 2) * (3 if px == 0 else 2)
            for dy, dx, ky, kx in zip(t.dy, t.dx, t.ky, t.kx):
                assert (2 * 0 + py) == 2 * dy - 2 + ky and (2 * 0 + px) == 2 * dx - 2 + kx    # oy = 2*iy - 2 + ky at y = 0
                assert (ky, kx) not in seen
                seen.add((ky, kx))
    assert len(seen) == 25
    t5 = E.conv_taps(5, 2)
    assert (min(t5.dy), max(t5.dy), len(t5.dy)) == (-2, 2, 25)


def test_warmup_cosine_lr_sequence_matches_reference():
    """GradualWarmupScheduler + CosineAnnealingLR stepped per epoch (TrainCondition.py:41-44, 70) vs the reference's sequence."""
    import warnings
    from hdiff_amd.Scheduler import GradualWarmupScheduler
    with open(os.path.join(GOLDEN, "lr_schedule.json")) as fh:
        ref = json.load(fh)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=ref["base_lr"], weight_decay=1e-4)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=opt, T_max=ref["epochs"], eta_min=0, last_epoch=-1)
    warm = GradualWarmupScheduler(optimizer=opt, multiplier=ref["multiplier"], warm_epoch=ref["epochs"] // 10,
                                  after_scheduler=cos)
    lrs = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(ref["epochs"]):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            warm.step()
    assert np.allclose(lrs, ref["lr_by_epoch"], rtol=1e-12, atol=0), max(abs(a - b) for a, b in zip(lrs, ref["lr_by_epoch"]))


def test_harness_helpers(tmp_path):
    from hdiff_amd.DiffusionFreeGuidence.TrainCondition import _epoch_indices
    from hdiff_amd.imageio import ImageDomainFolder, SyntheticDomains, save_image
    a, b = _epoch_indices(11, 3, 0, 2), _epoch_indices(11, 3, 1, 2)
    assert len(a) == len(b) == 6 and set(a) | set(b) == set(range(11))         # padded, strided, exhaustive
    assert _epoch_indices(11, 3, 0, 1) != _epoch_indices(11, 4, 0, 1)
    ds = SyntheticDomains(8, 16, 3)
    x, lab = ds[5]
    assert tuple(x.shape) == (3, 16, 16) and lab == 2 and float(x.abs().max()) <= 1.0
    assert torch.equal(ds[5][0], x)
    save_image(torch.rand(5, 3, 16, 16), str(tmp_path / "u" / "a.png"), nrow=4)
    save_image(torch.rand(3, 3, 16, 16), str(tmp_path / "w" / "b.png"), nrow=4)
    folder = ImageDomainFolder(str(tmp_path), 8)
    assert len(folder) == 2 and folder.domains == ["u", "w"]
    img, lab = folder[1]
    assert tuple(img.shape) == (3, 8, 8) and lab == 1 and -1.0 <= float(img.min()) and float(img.max()) <= 1.0


def test_psnr_ssim_definitions():
    """metrics.py against a direct evaluation of the published definitions (no skimage in this image)."""
    from hdiff_amd import metrics as M
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(24, 20, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, size=a.shape), 0, 255).astype(np.uint8)
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    assert abs(M.psnr(a, b, 255) - 10 * np.log10(255.0 ** 2 / mse)) < 1e-12
    assert M.psnr(a, a, 255) == float("inf") and abs(M.ssim(a, a, 255, channel_axis=2) - 1.0) < 1e-12
    # brute-force SSIM: every fully interior 7x7 window, sample covariance, per channel, then the mean
    vals = []
    x, y = a.astype(np.float64), b.astype(np.float64)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    for c in range(3):
        for i in range(3, 24 - 3):
            for j in range(3, 20 - 3):
                wx, wy = x[i - 3:i + 4, j - 3:j + 4, c].ravel(), y[i - 3:i + 4, j - 3:j + 4, c].ravel()
                ux, uy = wx.mean(), wy.mean()
                vx, vy = wx.var(ddof=1), wy.var(ddof=1)
                vxy = ((wx - ux) * (wy - uy)).sum() / 48.0
                vals.append(((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2)))
    assert abs(M.ssim(a, b, 255, channel_axis=2) - np.mean(vals)) < 1e-9
    p, s = M.batch_psnr_ssim(torch.rand(2, 3, 16, 16), torch.rand(2, 3, 16, 16))
    assert 0 < p < 20 and -1 <= s <= 1


def test_native_optimizer_is_a_torch_optimizer_and_refuses_cpu_parameters():
    """hdiff_amd.optim.AdamW (the clip + AdamW tail of a training step, csrc/optimizer.hip) without a GPU: constructs like torch's, takes the
    reference's schedulers, validates its hyper-parameters -- and raises on CPU parameters (no CPU path, like every operator of the package)."""
    from hdiff_amd import optim as HO
    from hdiff_amd.Scheduler import GradualWarmupScheduler
    p = torch.nn.Parameter(torch.randn(8))
    opt = HO.AdamW([p], lr=1e-4, weight_decay=1e-4)
    assert isinstance(opt, torch.optim.Optimizer) and opt.param_groups[0]["lr"] == 1e-4 and opt.param_groups[0]["weight_decay"] == 1e-4
    GradualWarmupScheduler(optimizer=opt, multiplier=2.5, warm_epoch=2,
                           after_scheduler=torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=opt, T_max=20, eta_min=0, last_epoch=-1))
    assert opt.step() is None                       # no gradients yet: nothing to do, like torch
    p.grad = torch.randn(8)
    with pytest.raises(RuntimeError, match="GPU"):
        opt.step(max_grad_norm=1.0)
    with pytest.raises(ValueError):
        HO.AdamW([p], lr=-1.0)
    with pytest.raises(ValueError):
        HO.AdamW([p], betas=(1.0, 0.999))


def test_ssim_psnr_hand_computed_single_window():
    """The hand-worked SSIM / PSNR values of tests/test_gpu_metrics.py (one 7x7 window, integers only) on the CPU suite too."""
    import test_gpu_metrics as G
    G.test_ssim_psnr_hand_computed_single_window()


def test_tree_b_state_dict_layout_and_seeded_init_match_reference():
    """DynamicUNet (diffusion/Model.py): the 319 keys / shapes of the reference, and the same weights from the same seed."""
    from hdiff_amd.diffusion.Model import DynamicUNet
    from hdiff_amd.diffusion.Diffusion import GaussianDiffusionSampler
    with open(os.path.join(GOLDEN, "state_dict_dyn_default.json")) as fh:
        ref = json.load(fh)
    d = np.load(os.path.join(GOLDEN, "dyn_unet_default64.npz"))
    cfg = json.loads(bytes(d["cfg_json"]).decode())
    torch.manual_seed(int(d["seed"][0]))
    m = DynamicUNet(**cfg)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == ref["entries"]
    assert sum(p.numel() for p in m.parameters()) == ref["n_params"] == 43237523
    sd = m.state_dict()
    names = sorted(sd.keys())
    assert names == list(d["weight_names"])
    for n, want in zip(names, d["weight_checksums"]):
        if n == "time_embedding.timembedding.0.weight":
            continue                      # sin/cos table: last-bit differences between CPU generations, pinned as data
        bits = sd[n].float().contiguous().reshape(-1).view(torch.int32).to(torch.int64)
        assert [int(bits.sum()), bits.numel(), int(bits[0]), int(bits[-1])] == list(want), n
    samp = GaussianDiffusionSampler(m, 1e-4, 0.02, 1000)
    assert [k for k in samp.state_dict() if not k.startswith("model.")] == ["betas", "coeff1", "coeff2", "posterior_var"]
    assert samp.alphas_bar.dtype == torch.float64 and torch.equal(samp.sqrt_alphas_bar, samp.alphas_bar)   # sic (reference :193)
    x = torch.zeros(1, 6, 16, 16)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        m.eval()(x, torch.zeros(1, dtype=torch.long))
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        samp(torch.zeros(1, 3, 16, 16))


@pytest.mark.parametrize("source,mnemonic", [("attention.hip", "v_pk_add_f32")])
def test_inline_asm_packed_math_keeps_the_trans_use_wait_state(tmp_path, source, mnemonic):
    """attention.hip sums softmax rows with v_pk_add_f32 written as inline asm.  gfx950 needs one wait state between a transcendental instruction (v_exp_f32 ...) and a VALU
    instruction that reads its result, and the compiler's hazard recogniser does not see inside asm statements -- the source
    pins all v_exp of a tile above the packed instructions.  This test compiles the file to ISA (no GPU needed) and checks
    that no such packed instruction reads a register written by the instruction directly in front of it when that one is a
    transcendental."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.isfile(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "hybrid-diffusion-underwater-atmopheric-image-enhancement_amd", "csrc")
    out = tmp_path / (source + ".s")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form",
                    "-S", "--cuda-device-only", os.path.join(csrc, source), "-o", str(out)], check=True, capture_output=True)

    def regs(tok):
        found = set()
        for m in re.finditer(r"v\[(\d+):(\d+)\]|v(\d+)", tok):
            found.update(range(int(m.group(1)), int(m.group(2)) + 1) if m.group(1) else [int(m.group(3))])
        return found
    ins = [l.strip() for l in out.read_text().split("\n")]
    ins = [l for l in ins if l and l[0] not in ";." and not l.endswith(":")]
    trans = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")
    n_pk = 0
    for prev, cur in zip(ins, ins[1:]):
        if cur.startswith(mnemonic):
            n_pk += 1
            if prev.startswith(trans):
                srcs = set().union(*[regs(o) for o in cur.split(None, 1)[1].split(",")[1:]])
                assert not (regs(prev.split(None, 1)[1].split(",")[0]) & srcs), (prev, cur)
    assert n_pk > 100            # the packed instructions are really there (the check above is not vacuous)


# ----------------------------------------------------------------------------------------------------------------------
# The drop-in for the reference's own entry files (north_star: "so MainCondition.py is a drop-in")
# ----------------------------------------------------------------------------------------------------------------------
# the reference's import statements, verbatim: MainCondition.py:1 and DiffusionFreeGuidence/TrainCondition.py:15-17
REFERENCE_IMPORT_LINES = (
    "from DiffusionFreeGuidence.TrainCondition import train, eval",
    "from DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler, GaussianDiffusionTrainer",
    "from DiffusionFreeGuidence.ModelCondition import UNet",
    "from Scheduler import GradualWarmupScheduler",
    # the second tree's (utils/rotinas.py:17-18)
    "from diffusion.Diffusion import GaussianDiffusionSampler as SamplerB",
    "from diffusion.Model import DynamicUNet",
)


def _run_fresh(code: str, cwd: str):
    return subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, cwd=cwd, timeout=300)


def test_install_dropin_serves_the_reference_import_lines(tmp_path):
    """A fresh interpreter, NOT started in the repo: install_dropin(), then the reference's import statements as they stand."""
    code = "\n".join([
        f"import sys; sys.path.insert(0, {ROOT!r})",
        "import hdiff_amd; mods = hdiff_amd.install_dropin()",
        *REFERENCE_IMPORT_LINES,
        "import MainCondition",
        "assert train.__module__ == 'hdiff_amd.DiffusionFreeGuidence.TrainCondition', train.__module__",
        "assert UNet is hdiff_amd.UNet and GaussianDiffusionSampler is hdiff_amd.GaussianDiffusionSampler",
        "assert GradualWarmupScheduler.__module__ == 'hdiff_amd.Scheduler'",
        "assert DynamicUNet.__module__ == 'hdiff_amd.diffusion.Model'",
        "assert MainCondition.main.__module__ == 'hdiff_amd.MainCondition'",
        "import DiffusionFreeGuidence as D; assert D.UNet is UNet and D.train is train",     # `from DiffusionFreeGuidence import *` users
        "print('OK', len(mods))",
    ])
    res = _run_fresh(code, str(tmp_path))
    assert res.returncode == 0 and "OK 9" in res.stdout, res.stderr[-3000:]


def test_install_dropin_refuses_to_shadow_an_imported_reference_module(tmp_path):
    (tmp_path / "Scheduler.py").write_text("X = 1\n")
    code = "\n".join([
        f"import sys; sys.path.insert(0, {ROOT!r})",
        "import Scheduler",                                  # somebody else's module of that name, imported first
        "import hdiff_amd",
        "try:\n    hdiff_amd.install_dropin()\nexcept ImportError as e:\n    print('REFUSED', e)",
        "assert 'DiffusionFreeGuidence' not in sys.modules",  # nothing half-installed
        "hdiff_amd.install_dropin(force=True); from Scheduler import GradualWarmupScheduler; print('FORCED')",
    ])
    res = _run_fresh(code, str(tmp_path))
    assert res.returncode == 0 and "REFUSED" in res.stdout and "FORCED" in res.stdout, res.stdout + res.stderr[-3000:]


REFERENCE_MAIN = "/root/reference/MainCondition.py"


@pytest.mark.skipif(not os.path.isfile(REFERENCE_MAIN), reason="the reference checkout exists in the build container only")
def test_reference_main_file_itself_runs_against_the_dropin(tmp_path):
    """The reference's REAL MainCondition.py, unmodified, executed from the reference's own directory (where its
    `DiffusionFreeGuidence/` package -- which does not even import, SyntaxError at ModelCondition.py:289 -- is first on
    sys.path): after install_dropin() its line 1 binds this package's train / eval and `main()` hands them the reference's
    default config.  train is replaced by a recorder (no GPU here); the config it receives must equal the defaults table
    of this package's MainCondition."""
    code = "\n".join([
        f"import sys, json, runpy; sys.path.insert(0, {ROOT!r})",
        "import hdiff_amd; hdiff_amd.install_dropin()",
        "import hdiff_amd.DiffusionFreeGuidence.TrainCondition as TC",
        "seen = []",
        "TC.train = lambda cfg: seen.append(('train', cfg))",
        "TC.eval = lambda cfg: seen.append(('eval', cfg))",
        f"ns = runpy.run_path({REFERENCE_MAIN!r}, run_name='__main__')",
        "assert len(seen) == 1 and seen[0][0] == 'train', seen",
        "from hdiff_amd.MainCondition import default_config",
        "assert seen[0][1] == default_config(), (seen[0][1], default_config())",
        "ns['main'](dict(default_config(), state='eval')); assert seen[1][0] == 'eval'",
        "print('OK')",
    ])
    res = _run_fresh(code, os.path.dirname(REFERENCE_MAIN))
    assert res.returncode == 0 and "OK" in res.stdout, res.stderr[-3000:]


def test_launch_ranks_relays_rank0_and_propagates_failure(tmp_path):
    """hdiff_amd.parallel.launch_ranks (what `bench.py --gpus N` becomes when no launcher started it): N fresh processes
    with the torch.distributed.run environment, rank 0's stdout relayed alone, a failing rank fails the job."""
    prog = tmp_path / "prog.py"
    prog.write_text("import os, sys\n"
                    "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                    "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                    "print('line from rank', r, 'of', w, sys.argv[1:])\n"
                    "sys.exit(3 if (len(sys.argv) > 2 and r == 1) else 0)\n")
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import hdiff_amd\n"
            "from hdiff_amd.parallel import launch_ranks\n"
            f"sys.exit(launch_ranks({str(prog)!r}, sys.argv[1:], 3))\n")
    ok = subprocess.run([os.sys.executable, "-c", code, "--x"], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stderr
    assert ok.stdout.strip().splitlines() == ["line from rank 0 of 3 ['--x']"]          # the other ranks' stdout went to stderr
    assert "line from rank 2 of 3" in ok.stderr
    bad = subprocess.run([os.sys.executable, "-c", code, "--x", "--fail"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 3 and "rank 1 exited with 3" in bad.stderr
    # the size it ships at (BASELINE configs C4 / C5: one node, 8 ranks)
    code8 = code.replace("sys.argv[1:], 3))", "sys.argv[1:], 8))")
    ok8 = subprocess.run([os.sys.executable, "-c", code8, "--x"], capture_output=True, text=True, timeout=180)
    assert ok8.returncode == 0, ok8.stderr
    assert ok8.stdout.strip().splitlines() == ["line from rank 0 of 8 ['--x']"]
    assert all(f"line from rank {r} of 8" in ok8.stderr for r in range(1, 8))
    bad8 = subprocess.run([os.sys.executable, "-c", code8, "--x", "--fail"], capture_output=True, text=True, timeout=180)
    assert bad8.returncode == 3 and "rank 1 exited with 3" in bad8.stderr


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """`bench.py --gpus 8` inside a 2-rank job (or the reverse) must not print a mislabelled line."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([os.sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True,
                         env=env, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr and not res.stdout.strip()


def test_roofline_traffic_table_is_stamped_and_goes_stale_with_the_kernel_sources(tmp_path, monkeypatch):
    """profiles/roofline_traffic.json carries the commit it was measured at and a hash of each entry's kernel sources;
    bench.load_traffic() reports an entry only while the sources' CODE (comments and whitespace stripped) still hashes to that (a
    stale counter is not a measurement of the run that prints it)."""
    import bench
    table, stamp, issue = bench.load_traffic()
    assert set(issue) <= set(table)                      # instruction counts are reported only beside fresh traffic entries
    assert stamp.get("measured_at_commit") and stamp.get("measured_utc")
    assert set(stamp["kernels"]) >= {"mha_flash_fwd_L65536_B16", "mha_flash_fwd_L65536_B16_bf16x3", "gn_stats_128_256_B16"}
    for key, state in stamp["kernels"].items():
        assert (state == "fresh") == (key in table), (key, state)
    # a changed kernel source: its entries disappear from the table, the others stay
    import shutil
    fake = tmp_path / "csrc"
    shutil.copytree(bench.CSRC, fake, ignore=shutil.ignore_patterns("build"))
    # (the stamp hashes CODE: a comment or a blank line leaves an entry fresh ...)
    with open(fake / "attention.hip", "a") as fh:
        fh.write("// edited\n\n/* a block\n   comment */\n")
    monkeypatch.setattr(bench, "CSRC", str(fake))
    table_c, stamp_c, _ = bench.load_traffic()
    assert stamp_c["kernels"]["mha_flash_fwd_L65536_B16"] == stamp["kernels"]["mha_flash_fwd_L65536_B16"]
    # (... a changed token does not)
    with open(fake / "attention.hip", "a") as fh:
        fh.write("static int edited_marker = 1;\n")
    table2, stamp2, issue2 = bench.load_traffic()
    assert "mha_flash_fwd_L65536_B16" not in issue2
    assert "mha_flash_fwd_L65536_B16" not in table2 and stamp2["kernels"]["mha_flash_fwd_L65536_B16"].startswith("STALE")
    if stamp["kernels"]["gn_stats_128_256_B16"] == "fresh":
        assert "gn_stats_128_256_B16" in table2
