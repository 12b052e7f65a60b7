#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec of the CFG-DDPM sampler at 256x256 on the T=1000 schedule.

  python bench.py --gpus N --steps K --warmup W

N > 1 with no launcher: this process (which never touches the GPU) starts N fresh rank processes itself
(hdiff_amd.parallel.launch_ranks -- the reference's other tree does the same with mp.spawn, utils/rotinas.py:572-577),
waits, relays rank 0's ONE JSON line and exits non-zero if any rank does.  Under ``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`` the ranks are already there (WORLD_SIZE is set) and each runs its share.
Either way one rank per GPU over RCCL, and the run refuses to time anything unless WORLD_SIZE == --gpus.

One "step" = one iteration of the reference sampler loop (DiffusionCondition.py:87-96) for the whole per-GPU batch:
cond + uncond UNet forward (one 2B-batched launch sequence) + fused CFG/posterior update.  Inputs (x_T, labels, weights)
are resident in HBM before the timed region.  Sampling shards by image: every rank runs its own batch with its own seed
and there is no data-path collective (weak scaling); value = N * K / max-over-ranks wall time.

The timed region replays the hipGraph-captured step (the north-star formulation; ``--eager`` times plain launches instead).
Per-kernel durations come from a second pass of the same K steps (same start state, same time steps) issued as plain
launches with HIP events recorded on the launch stream around the launches of interest (events cannot bracket nodes
inside a graph replay; the kernels, their arguments and their order are identical -- the two passes agree to < 0.1 % at
this size, both wall times are reported).

Rank 0 prints ONE JSON line with the contract keys plus
  roofline      dominant kernel (flash attention at L = 65536, d_head = 16): ``frac`` against the 16-bit dense MFMA peak / the
                3.5 piece products it executes per fp32 product (fp32-MFMA peak in the f32 mode), ``frac_vs_16bit_peak`` against
                the raw peak, ``issue_model_ms`` / ``frac_vs_issue_model`` (its own PMC instruction counts priced at issue cost at
                the clock the run held), ``matrix_only_floor_ms_at_power_limit`` / ``frac_vs_matrix_only_floor`` (its MFMA count at
                the rate a bare MFMA loop sustains on this board: the kernel is bound by the energy of its instruction stream at
                the board's power limit, ``bound_note``); ``secondary`` holds the full-resolution 3x3 convolution (MFMA) and the
                GroupNorm statistics pass (HBM); ``traffic`` and the instruction counts come from profiles/roofline_traffic.json
                and are reported only while the kernel sources still hash to what that file was measured on (``traffic_stamp``)
  device_clock / box_range   engine clock and board power of THIS run's device during the timed region; the range of the headline
                over the boxes met (the power limit leaves different clocks on different boxes)
  cpu_baseline  the CPU oracle (oracle/cpu_path.py, kind "port"): one whole UNet forward at 128x128 timed on the host
                cores (a bounded sample), scaled to the benchmark's step by the algorithmic FLOP ratio; ``same_config``
                holds the one BASELINE config the CPU finishes in full (C1), timed on both sides
  parity        max-abs / PSNR of the HIP path against the oracle on the inputs the two legs above computed anyway, and
                `modes`: the sampler state of the full-size workload after the same steps in the two contraction modes
  configs       the other BASELINE configs on this GPU (N = 1 only; never part of ``value``): C1 in full, C2 20 steps,
                C5 one step of one GPU's share, C3 one optimizer step (``--no-extras`` skips them)
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402   (importing torch does not initialise the GPU)

MODEL = dict(T=1000, num_labels=10, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.15)
BETA = (1e-4, 0.02)
GUIDANCE_W = 1.8
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2516.6    # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
MFMA16_NS_AT_POWER_LIMIT = 8.1    # tools/mfma_k16_probe.hip (profiles/r06_attention_bwd_pipeline.txt (5)): a bare v_mfma_f32_16x16x32_f16 loop, two waves
                                  # per SIMD, is clocked down to ~1.99 GHz by the board's power limit: 16.1 cycles = 8.1 ns per MFMA per SIMD
N_SIMD = 1024                     # 256 CUs x 4
# The headline is ONE sample of a box-to-box range (same build, different boxes of the pool: the kernel sits on the board's power limit and boxes
# hold different clocks there).  Measured ranges of the round, updated with the round's records:
BOX_RANGE = {"denoising_steps_per_s": [2.76, 3.00], "sclk_mhz_mean": [2159, 2315],      # this round's two driver-shaped runs: 2.866 @ 2199 MHz, 2.919 @ 2254 MHz "board_power_w_mean": [1301, 1332],
             "what": "256x256, batch 8 headline on the boxes met in rounds 5-6 (lowest = a heat-soaked device right after the GPU suite)",
             "source": "profiles/r05_bench_full_line*.json, profiles/r06_bench_full_line*.json"}
SCLK_MAX_MHZ = 2400.0             # MI355X_MICROARCH.md: engine clock ceiling
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s is what a float4 copy achieves
# algorithmic FLOPs per sample per UNet forward (torch flop counter on the reference UNet)
FWD_GFLOP = {64: 74.0, 128: 529.6, 256: 5857.4, 512: 83254.1}
CSRC = os.path.join(ROOT, "hybrid-diffusion-underwater-atmopheric-image-enhancement_amd", "csrc")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contract", choices=["f32", "bf16x3"], default="bf16x3",
                    help="attention / 3x3-conv contractions: the fp32-class three-piece bf16 split on the bf16 MFMA (the library's "
                         "default) or the fp32-input MFMA; the other one is run once beside the headline (`other_contract_mode`)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra (untimed in `value`) run in the other contraction mode")
    ap.add_argument("--eager", action="store_true", help="time plain launches instead of the hipGraph replay")
    ap.add_argument("--graph", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the second, event-instrumented pass (no roofline)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the other BASELINE configs (C1 full on GPU and CPU, C2, C5, C3) reported under `configs`")
    ap.add_argument("--extras-scale", choices=["full", "small"], default="full",
                    help="dev/tests: `small` runs the `configs` legs at toy sizes so that the code path finishes in seconds")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="dev: run the N > 1 code path (rendezvous, barriers, max over ranks, aggregation) with every rank on "
                         "cuda:0 over gloo -- RCCL refuses two ranks on one device; the numbers mean nothing")
    a = ap.parse_args(argv)
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    # three passes (timed, event-instrumented, other contraction mode) each start from step T-1: K + W steps must fit
    if a.steps + a.warmup + 1 > MODEL["T"]:
        ap.error(f"--steps + --warmup must stay below T = {MODEL['T']}")
    return a


# ----------------------------------------------------------------------------------------------------------------------
# helpers shared by the headline and the `configs` legs
# ----------------------------------------------------------------------------------------------------------------------
def _model(cfg, dev, train=False):
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    torch.manual_seed(0)                              # same weights on every rank (replicated model) and on the CPU side
    m = UNet(**cfg)
    return (m.train() if train else m.eval()).to(dev)


def _oracle_cfg(cfg):
    from oracle import cpu_path as O
    return O.UNetConfig(T=cfg["T"], num_labels=cfg["num_labels"], ch=cfg["ch"], ch_mult=tuple(cfg["ch_mult"]),
                        num_res_blocks=cfg["num_res_blocks"])


def _host_threads():
    threads = min(os.cpu_count() or 1, 16)            # the one-GPU box's CPU share
    torch.set_num_threads(threads)
    return threads


def _quality(a, b):
    """PSNR (dB) / SSIM of two [-1, 1] image batches on the reference's evaluation scale (x*0.5+0.5, utils/rotinas.py:922-926)."""
    from hdiff_amd import metrics as M
    ia = ((a.cpu().clamp(-1, 1) * 0.5 + 0.5) * 255.0).permute(0, 2, 3, 1).numpy()
    ib = ((b.cpu().clamp(-1, 1) * 0.5 + 0.5) * 255.0).permute(0, 2, 3, 1).numpy()
    return (min(M.psnr(x, y, 255) for x, y in zip(ia, ib)),
            min(M.ssim(x, y, 255, channel_axis=2) for x, y in zip(ia, ib)))


class ClockSampler:
    """Samples the device's engine clock (and board power, where the driver exposes it) from sysfs while the timed region
    runs: the split-operand attention kernels are sensitive to the clock a device holds under their MFMA + vector mix, and
    boxes differ by up to 9 % on the same build (round 3) -- the line then says what clock its number was measured at.
    A reading thread at 20 Hz on files; nothing is launched on the GPU and nothing runs if the files are not there."""

    def __init__(self, device_index=0):
        import glob
        self.freq, self.power, self.dpm, self.card = None, None, None, None
        cards = sorted(glob.glob("/sys/class/drm/card*/device"))
        amd = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
        # the sysfs card of THIS process's device: by PCI address (a box shows every GPU of its host under /sys, the process
        # sees one of them -- card0 is usually somebody else's, idle, GPU)
        want = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            pass
        pick = [c for c in amd if want and os.path.basename(os.path.realpath(c)) == want]
        if not pick and len(amd) == 1:
            pick = amd
        if pick:
            self.card = pick[0]
            for h in sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*"))):
                for fi in sorted(glob.glob(os.path.join(h, "freq*_input"))):
                    lab = (self._read(fi.replace("_input", "_label")) or "").strip()
                    if lab == "sclk" or (self.freq is None and lab == ""):
                        self.freq = fi
                for pn in ("power1_average", "power1_input"):
                    if os.path.exists(os.path.join(h, pn)):
                        self.power = os.path.join(h, pn)
                        break
            self.dpm = os.path.join(self.card, "pp_dpm_sclk")
        self.want = want
        self.mhz, self.watts, self._stop, self._thread = [], [], False, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except OSError:
            return None

    def _once(self):
        v = self._read(self.freq) if self.freq else None
        if v and v.strip().isdigit():
            self.mhz.append(int(v) / 1e6)
        elif self.dpm:
            txt = self._read(self.dpm) or ""
            for line in txt.splitlines():
                if line.strip().endswith("*"):
                    try:
                        self.mhz.append(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
                    except (IndexError, ValueError):
                        pass
        w = self._read(self.power) if self.power else None
        if w and w.strip().isdigit():
            self.watts.append(int(w) / 1e6)

    def __enter__(self):
        import threading

        def loop():
            while not self._stop:
                self._once()
                time.sleep(0.05)
        if self.freq or self.dpm:
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join(1.0)

    def summary(self):
        if not self.mhz:
            return {"sclk_mhz_mean": None, "note": f"no readable engine-clock file for PCI device {self.want} under /sys/class/drm/card*/device "
                                                   "(hwmon freq*_input labelled sclk / pp_dpm_sclk)"}
        out = {"sclk_mhz_mean": round(sum(self.mhz) / len(self.mhz), 1), "sclk_mhz_min": min(self.mhz), "sclk_mhz_max": max(self.mhz),
               "samples": len(self.mhz), "source": self.freq or self.dpm, "pci": self.want, "sampled": "during the timed region, 20 Hz"}
        if self.watts:
            out["board_power_w_mean"] = round(sum(self.watts) / len(self.watts), 1)
        return out


def _release():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def time_sampler_steps(model, size, batch, T, beta, w, steps, warmup, dev, seed=1234, graph=True):
    """K replays of the captured denoising step at (size, batch) after W warm-up replays; returns seconds per step."""
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler, _SamplerPlan
    sampler = GaussianDiffusionSampler(model, beta[0], beta[1], T, w=w).to(dev)
    g = torch.Generator().manual_seed(seed)
    x_T = torch.randn(batch, 3, size, size, generator=g).to(dev)
    labels = (torch.arange(batch) % 2 + 1).to(dev)
    with torch.no_grad():
        sp = _SamplerPlan(sampler, batch, size, size, dev)
        plan = sp.variant(False, seed)
        sp.unet.plan.pack_weights()
        sp.reset(x_T, labels)
        if graph:
            plan.capture()
        run = plan.replay if graph else plan.run
        for _ in range(warmup):
            run()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        assert int(sp.nan_flag.item()) == 0, "nan in tensor."
        mem = sp.unet.plan.bytes_allocated()
    del plan, sp, sampler
    return dt, mem


def train_steps(size, batch, steps, warmup, dropout, dev, rank=0, world=1, rehearse=False, model_cfg=None):
    """Optimizer steps of the CFG-DDPM trainer on the HIP path: forward + hand-written backward [+ the one gradient exchange
    of data-parallel training] + clip_grad_norm_ + AdamW (TrainCondition.py:59-63).  Returns a dict for the JSON line."""
    from hdiff_amd import parallel
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionTrainer
    m = _model(dict(model_cfg or MODEL, dropout=dropout), dev, train=True)
    parallel.broadcast_parameters_(m.parameters())
    tr = GaussianDiffusionTrainer(m, BETA[0], BETA[1], MODEL["T"]).to(dev)
    weights = list(m.parameters())
    from hdiff_amd import optim as hdiff_optim
    opt = hdiff_optim.AdamW(weights, lr=1e-4, weight_decay=1e-4)      # the native tail: clip + AdamW in three launches (csrc/optimizer.hip)
    flat = parallel.FlatGradients(weights, world, overlap=True) if world > 1 else None
    g = torch.Generator().manual_seed(1 + rank)              # per-rank data
    x0 = (torch.rand(batch, 3, size, size, generator=g) * 2 - 1).to(dev)
    labels = (torch.arange(batch) % 2 + 1).to(dev)
    torch.manual_seed(100 + rank)                            # per-rank t / noise / dropout streams
    exchanged = 0
    loss = None

    def step():
        nonlocal exchanged, loss
        if flat is not None:
            flat.zero_()
        else:
            opt.zero_grad()
        loss = tr(x0, labels).sum() / batch ** 2.
        loss.backward()
        if flat is not None:
            exchanged = flat.exchange_mean_()
        return opt.step(max_grad_norm=1.0)                        # clip_grad_norm_(weights, 1.0); AdamW.step()

    torch.cuda.reset_peak_memory_stats(dev)
    for _ in range(warmup):
        step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        gn = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    dt = parallel.max_over_ranks((time.perf_counter() - t0) / steps, None if rehearse else dev)
    fwd = FWD_GFLOP.get(size, FWD_GFLOP[64] * (size / 64) ** 2)
    res = {"size": size, "batch_per_gpu": batch, "n_gpus": world, "s_per_step": dt, "samples_per_s": world * batch / dt,
           "fwd_equiv_tflops_per_gpu": 3 * fwd * batch / 1e3 / dt, "loss": float(loss.detach()), "grad_norm": float(gn),
           "all_grads_present": all(p.grad is not None for p in weights),
           "max_mem_GB": torch.cuda.max_memory_allocated(dev) / 1e9, "gradient_exchange_bytes_per_rank": exchanged,
           "dropout": dropout, "steps": steps, "warmup": warmup}
    del opt, tr, m, weights, flat
    return res


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline + parity
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(size, batch, dev):
    """Oracle on the host cores: ONE whole UNet forward at 128x128, B = 1 (the largest size that finishes in seconds;
    attention is 60 % of its FLOPs against 85 % at 256x256), scaled to one denoising step of the benchmark (2 * batch
    forwards at `size`) by the algorithmic FLOP ratio.  The same input then goes through the HIP path: `parity`."""
    from oracle import cpu_path as O
    threads = _host_threads()
    m = _model(MODEL, "cpu")
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    cfg = _oracle_cfg(MODEL)
    S = 128
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, S, S, generator=g)
    t, lab = torch.tensor([500]), torch.tensor([1])
    with torch.no_grad():
        O.unet_forward(sd, cfg, torch.randn(1, 3, 32, 32, generator=g), torch.tensor([5]), torch.tensor([1]))  # warm
        t0 = time.perf_counter()
        ref = O.unet_forward(sd, cfg, x, t, lab)
        dt = time.perf_counter() - t0
        md = m.to(dev)
        eps = md(x.to(dev), t.to(dev), lab.to(dev)).cpu()
    del md, m
    err = (eps - ref).abs().max().item()
    rng = (ref.max() - ref.min()).item()
    mse = ((eps.double() - ref.double()) ** 2).mean().item()
    parity = {"what": f"UNet forward (default model, seed-0 init) at {S}x{S}, B=1, t=500, label=1: HIP path vs CPU oracle "
                      "(oracle/cpu_path.py, pinned to the real reference by tests/golden/unet_default128.npz)",
              "max_abs": err, "ref_abs_max": ref.abs().max().item(),
              "psnr_db": float("inf") if mse == 0 else 10.0 * __import__("math").log10(rng * rng / mse),
              "psnr_note": "on eps with data_range = max - min of the oracle's output", "tolerance_max_abs": 2e-4}
    fwd = FWD_GFLOP[size] if size in FWD_GFLOP else FWD_GFLOP[S] * (size / S) ** 4
    scale = 2 * batch * fwd / FWD_GFLOP[S]
    base = {"value": 1.0 / (dt * scale), "unit": "denoising-steps/s", "cores": threads, "kind": "port",
            "sample": f"one UNet forward at {S}x{S}, B=1 on the CPU oracle (blockwise attention) took {dt:.1f} s = "
                      f"{FWD_GFLOP[S] / dt:.0f} GFLOP/s; one step of the benchmark is 2x{batch} forwards at {size}x{size} = "
                      f"{scale:.0f}x its FLOPs (the reference's own materialised-attention formulation cannot run at "
                      "256x256: 137 GB/sample)",
            "measured_s": dt, "torch_threads": threads}
    return base, parity


def config_c1(dev, small=False):
    """BASELINE config C1 in full on both sides: 64x64, T=50, B=1, w=1.8, default UNet -- the HIP sampler (hipGraph replay)
    and the CPU oracle (timed in full on the host cores), shared weights / x_T / per-step noise -> the real same-config
    GPU/CPU ratio and the end-to-end PSNR / SSIM the metric asks for."""
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler
    from oracle import cpu_path as O
    T_, S, w, beta = (4, 32, 1.8, (1e-4, 0.028)) if small else (50, 64, 1.8, (1e-4, 0.028))
    cfg = dict(MODEL, T=T_) if not small else dict(T=T_, num_labels=10, ch=32, ch_mult=[1, 2], num_res_blocks=1, dropout=0.0)
    m = _model(cfg, "cpu")
    with torch.no_grad():
        m.tail[2].weight.mul_(0.1)       # default init saturates x to +-1 within a few steps; keep x O(1)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    x_T = torch.randn(1, 3, S, S, generator=g)
    labels = torch.tensor([1])
    noise = torch.randn(T_, 1, 3, S, S, generator=g)
    threads = _host_threads()
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = O.sampler_forward(sd, _oracle_cfg(cfg), beta[0], beta[1], T_, w, x_T, labels, noise)
        cpu_s = time.perf_counter() - t0
        md = m.to(dev)
        samp = GaussianDiffusionSampler(md, beta[0], beta[1], T_, w=w).to(dev)
        out = samp(x_T.to(dev), labels.to(dev), noise_by_step=noise.to(dev))            # parity run (injected noise)
        samp(x_T.to(dev), labels.to(dev))                                                 # warm: plan, pack, capture
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        samp(x_T.to(dev), labels.to(dev))                                                 # the reference call, whole
        torch.cuda.synchronize(dev)
        gpu_s = time.perf_counter() - t0
    psnr, ssim = _quality(out, ref)
    res = {"workload": f"C1: sampling {S}x{S}, T={T_}, B=1, w={w}, beta {beta}; GaussianDiffusionSampler.forward whole "
                       "(pack + capture + T graph replays + clip)",
           "gpu_steps_per_s": T_ / gpu_s, "gpu_s": gpu_s, "cpu_steps_per_s": T_ / cpu_s, "cpu_s": cpu_s,
           "cpu_cores": threads, "cpu_kind": "port (oracle/cpu_path.py, all T steps timed)", "gpu_over_cpu": cpu_s / gpu_s,
           "parity": {"max_abs": (out.cpu() - ref).abs().max().item(), "psnr_db": psnr, "ssim": ssim,
                      "unsaturated_fraction": (ref.abs() < 1.0).float().mean().item(),
                      "note": "shared weights (tail conv scaled by 0.1 so the trajectory does not saturate), x_T and "
                              "per-step noise; PSNR/SSIM on x*0.5+0.5 at 8-bit range like utils/rotinas.py:922-926"}}
    del samp, md, m
    return res


def other_configs(dev, small=False):
    """BASELINE configs C1, C2, C5 (one GPU's share) and C3 on this GPU.  Never part of `value`."""
    out = {}
    t_all = time.perf_counter()
    out["C1"] = config_c1(dev, small)
    _release()
    mcfg = MODEL if not small else dict(T=1000, num_labels=10, ch=32, ch_mult=[1, 2], num_res_blocks=1, dropout=0.0)
    model = _model(mcfg, dev)
    S2, B2, K2 = (32, 2, 3) if small else (128, 16, 20)
    dt, mem = time_sampler_steps(model, S2, B2, 200, (1e-4, 0.028), GUIDANCE_W, K2, 2, dev)
    out["C2"] = {"workload": f"C2: sampling {S2}x{S2}, T=200 schedule, B={B2}, w={GUIDANCE_W}, hipGraph replay, {K2} steps "
                             "after 2 warm-up steps",
                 "steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "sample_steps_per_s": B2 / dt,
                 "whole_step_tflops": None if small else 2 * B2 * FWD_GFLOP[S2] / 1e3 / dt, "plan_bytes": mem}
    _release()
    S5, B5 = (64, 2) if small else (512, 8)
    dt, mem = time_sampler_steps(model, S5, B5, MODEL["T"], BETA, GUIDANCE_W, 1, 1, dev)
    out["C5"] = {"workload": f"C5 (one GPU's share): sampling {S5}x{S5}, T=1000 schedule, B={B5}/GPU, w={GUIDANCE_W}, hipGraph "
                             "replay, 1 step after 1 warm-up step",
                 "steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "sample_steps_per_s": B5 / dt,
                 "whole_step_tflops": None if small else 2 * B5 * FWD_GFLOP[S5] / 1e3 / dt, "plan_bytes": mem}
    del model
    _release()
    S3, B3 = (32, 2) if small else (256, 64)
    r = train_steps(S3, B3, 1, 1, 0.15, dev, model_cfg=mcfg)
    r["workload"] = (f"C3: training {S3}x{S3}, T=1000, B={B3}, dropout 0.15: one optimizer step (forward + hand-written "
                     "backward + clip_grad_norm_ + AdamW, TrainCondition.py:59-63) after 1 warm-up step")
    out["C3"] = r
    _release()
    out["seconds_spent"] = time.perf_counter() - t_all
    return out


# ----------------------------------------------------------------------------------------------------------------------
# roofline.traffic: counter values measured by tools/profile_round.sh, valid only for the kernel sources they were taken on
# ----------------------------------------------------------------------------------------------------------------------
def _code_only(text):
    """C++ source without comments and without whitespace: what the stamp of a traffic entry hashes, so that an edit of a comment
    does not make a measurement look stale (string literals of these files hold no comment markers)."""
    import re
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


def source_hash(names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), "r", encoding="utf-8", errors="replace") as f:
            h.update(n.encode() + b"\0" + _code_only(f.read()).encode())
    return h.hexdigest()[:16]


def load_traffic():
    """(table, stamp, issue counts): entries of profiles/roofline_traffic.json whose kernel sources still hash to what they were measured on."""
    prof = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        tab = json.load(open(prof))
    except Exception:
        return {}, {"status": "absent"}, {}
    stamp = tab.get("_stamp") or {}
    out, status = {}, {}
    for key, meta in (stamp.get("kernels") or {}).items():
        try:
            now = source_hash(meta["sources"])
        except OSError:
            now = None
        fresh = now == meta.get("sha256_16")
        status[key] = "fresh" if fresh else "STALE (kernel source changed since the counters were collected): not reported"
        if fresh and key in tab:
            out[key] = tab[key]
    issue = {k: v for k, v in (tab.get("_issue_counts") or {}).items() if k in out}      # instruction counts: only beside fresh traffic entries
    return out, {"measured_at_commit": stamp.get("commit"), "measured_utc": stamp.get("utc"),
                 "collected_by": stamp.get("how"), "kernels": status}, issue


# ----------------------------------------------------------------------------------------------------------------------
def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain program: become the launcher.  Nothing above has touched the GPU.
        from hdiff_amd.parallel import launch_ranks
        sys.exit(launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: refusing to label a {world}-rank run as {a.gpus} GPUs")
    dist = world > 1
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.rehearse_one_gpu:
            local = 0
            torch.cuda.set_device(0)
            td.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local)
            td.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if dist else 0)

    import hdiff_amd
    from hdiff_amd import _capi
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler, _SamplerPlan
    lib = hdiff_amd.lib()

    S, B, K, Wm = a.size, a.batch, a.steps, a.warmup
    hdiff_amd.set_contraction_mode(a.contract)
    model = _model(MODEL, dev)
    sampler = GaussianDiffusionSampler(model, BETA[0], BETA[1], MODEL["T"], w=GUIDANCE_W).to(dev)
    g = torch.Generator().manual_seed(1234 + rank)    # per-rank data
    x_T = torch.randn(B, 3, S, S, generator=g).to(dev)
    labels = (torch.arange(B) % 2 + 1).to(dev)

    with torch.no_grad():
        sp = _SamplerPlan(sampler, B, S, S, dev)
        plan = sp.variant(False, 1234 + rank)
        sp.unet.plan.pack_weights()
        stream = torch.cuda.current_stream(dev).cuda_stream
        use_graph = not a.eager
        if use_graph:
            plan.capture()

        def reset():
            """Every pass starts from the same state: x = x_T at time step T-1 (so the passes run the same time steps and the
            step counter never leaves the schedule, whatever K is)."""
            sp.reset(x_T, labels)

        # launches of interest: attention over the full-resolution token set (dominant), the full-resolution 128 -> 128 3x3
        # convolutions, the full-resolution GroupNorm statistics passes
        L_full = S * S
        Cc = MODEL["ch"] * MODEL["ch_mult"][0]
        groups = {"attn": [], "conv3x3": [], "gn_stats": []}
        for i, (name, _, args) in enumerate(plan.ops):
            if name in ("hdiff_mha_flash_fwd", "hdiff_mha_flash_fwd_ws") and args[6] == L_full:
                groups["attn"].append(i)
            elif name == "hdiff_conv2d_fwd":
                d = args[0]._obj
                if d.ntaps == 9 and d.C0 == Cc and d.C1 == 0 and d.Cout == Cc and d.H == S and d.in_stride == 1:
                    groups["conv3x3"].append(i)
            elif name in ("hdiff_gn_stats", "hdiff_gn_scale_shift") and args[5] == L_full and args[2] + args[3] == Cc:
                groups["gn_stats"].append(i)
        hooked = {i: g for g, idx in groups.items() for i in idx}
        events = {g: [] for g in groups}

        def one_step(instrumented=False):
            if use_graph and not instrumented:
                plan.replay()
                return
            for i, (name, fn, args) in enumerate(plan.ops):
                hook = instrumented and i in hooked
                if hook:
                    e0, e1 = C.c_void_p(), C.c_void_p()
                    lib.hdiff_event_create(C.byref(e0)); lib.hdiff_event_create(C.byref(e1))
                    lib.hdiff_event_record(e0, stream)
                rc = fn(*args, stream)
                if rc != 0:
                    _capi.check(rc, name)
                if hook:
                    lib.hdiff_event_record(e1, stream)
                    events[hooked[i]].append((e0, e1))

        reset()
        for _ in range(Wm):
            one_step()
        if dist:
            td.barrier()
        torch.cuda.synchronize(dev)
        clock = ClockSampler(int(os.environ.get("LOCAL_RANK", "0")) if not a.rehearse_one_gpu else 0)
        with clock:
            t0 = time.perf_counter()
            for _ in range(K):
                one_step()
            torch.cuda.synchronize(dev)
            if dist:
                td.barrier()
            elapsed = time.perf_counter() - t0
        assert int(sp.nan_flag.item()) == 0, "nan in tensor."
        assert int(sp.step.item()) == MODEL["T"] - 1 - (K + Wm)
        x_timed = sp.x.clone()                         # the state after Wm + K steps in the timed mode (for `parity.modes`)

        # second pass: the same steps (same start state, same time steps) as plain launches with HIP events around the
        # launches of interest; the warm-up steps are replayed so that the K instrumented steps are the K timed ones
        kernel_pass_s = None
        if not a.no_kernel_pass:
            reset()
            for _ in range(Wm):
                one_step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(K):
                one_step(instrumented=True)
            torch.cuda.synchronize(dev)
            kernel_pass_s = (time.perf_counter() - t1) / K
            assert int(sp.nan_flag.item()) == 0, "nan in tensor."

        # the other contraction mode, reported beside the headline (never part of `value`)
        alt = None
        mode_parity = None
        if world == 1 and not a.no_alt:
            other = "bf16x3" if a.contract == "f32" else "f32"
            hdiff_amd.set_contraction_mode(other)
            use_graph = False                         # a captured graph bakes the contraction mode: plain launches here
            reset()
            for _ in range(Wm):                       # the same Wm + K steps (same x_T, same Philox noise) as the timed pass
                one_step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(K):
                one_step()
            torch.cuda.synchronize(dev)
            dt_alt = (time.perf_counter() - t1) / K
            hdiff_amd.set_contraction_mode(a.contract)
            assert int(sp.nan_flag.item()) == 0, "nan in tensor."
            diff = (sp.x.double() - x_timed.double()).abs()
            mode_parity = {"what": f"pre-clip sampler state after {Wm + K} denoising steps of the benchmark workload ({S}x{S}, batch "
                                   f"{B}, same x_T and in-kernel noise) in the two contraction modes, {a.contract} vs {other}",
                           "max_abs": diff.max().item(), "rms": diff.pow(2).mean().sqrt().item(),
                           "state_abs_max": x_timed.abs().max().item(), "state_rms": x_timed.double().pow(2).mean().sqrt().item()}
            alt = {"contract": other, "ms_per_step": dt_alt * 1e3, "denoising_steps_per_s": 1.0 / dt_alt,
                   "note": "same workload with the attention / 3x3-conv contractions in the other mode (plain launches); "
                           "f32 = fp32-input MFMA (exact k-ordered fma chain); bf16x3 = the split-operand mode (the name is "
                           "round 3's): every fp32 operand as 16-bit pieces -- fp16 pairs where its range is known or "
                           "balanced per product term, bf16 triples elsewhere --, products on the 16-bit MFMA, fp32 "
                           "accumulate (fp32-class error, tests/test_gpu_ops.py::test_flash_attention_split_bf16_is_fp32_class)"}

    if dist:
        tmax = torch.tensor([elapsed], device="cpu" if a.rehearse_one_gpu else dev, dtype=torch.float64)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax.item())

    durations = {}
    for gname, evs in events.items():
        vals = []
        for e0, e1 in evs:
            ms = C.c_float()
            lib.hdiff_event_elapsed_ms(e0, e1, C.byref(ms))
            vals.append(ms.value)
            lib.hdiff_event_destroy(e0); lib.hdiff_event_destroy(e1)
        durations[gname] = vals

    if rank == 0:
        traffic_tab, traffic_stamp, issue_tab = load_traffic()
        flops_per_launch = 4.0 * L_full * L_full * Cc * (2 * B)          # QK^T + PV over 8 heads, 2B samples (CFG)
        roof = None
        att_ms = durations.get("attn") or []
        if att_ms:
            avg = sum(att_ms) / len(att_ms)
            ach = flops_per_launch / (avg * 1e-3) / 1e12
            if a.contract == "f32":
                kname, peak = "mha_flash_fwd_fast_kernel<16,4>", PEAK_F32_MFMA_TFLOPS
            else:
                # The executed 16-bit products per fp32 product set the scheme's fp32-equivalent matrix ceiling.  Round 5: Q K^T as
                # fp16 pairs with a balance per product term = four, P V as fp16 pairs = three (attention_h2.hip) -- 3.5 on average
                # over the launch's two equal halves.  (Round 4: six bf16-triple products for Q K^T, 4.5 on average, peak 559.2;
                # round 3: six for both, peak 419.4 -- `frac_vs_peak_div4_5` / `frac_vs_peak_div6` keep those denominators for
                # comparison across rounds.  The kernel sits on the board's POWER limit, not on a pipe: DESIGN.md section 4.)
                kname, peak = ("split-operand flash kernel mha_flash_fwd_h2_kernel (attention_h2.hip: Q K^T as four fp16 piece "
                               "products with a balance per term, P V as three fp16 piece products, fp32 accumulate; peak = 16-bit "
                               "dense MFMA peak / 3.5 executed products per fp32 product)"), PEAK_BF16_MFMA_TFLOPS / 3.5
            roof = {"bound": "mfma",
                    "kernel": f"hdiff_mha_flash_fwd = {kname} + overflow-check pass, L={L_full} d_head=16 "
                              f"heads=8 batch={2 * B}",
                    "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4),
                    "frac_vs_16bit_peak": None if a.contract == "f32" else round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                    "frac_vs_peak_div4_5": None if a.contract == "f32" else round(ach / (PEAK_BF16_MFMA_TFLOPS / 4.5), 4),
                    "frac_vs_peak_div6": None if a.contract == "f32" else round(ach / (PEAK_BF16_MFMA_TFLOPS / 6), 4),
                    "traffic": traffic_tab.get(f"mha_flash_fwd_L{L_full}_B{2 * B}" + ("" if a.contract == "f32" else "_bf16x3")),
                    "traffic_stamp": traffic_stamp,
                    "avg_launch_ms": round(avg, 3), "launches_timed": len(att_ms),
                    "algorithmic_flop_per_launch": flops_per_launch,
                    "timed_in": "second pass of the same K steps as plain launches (HIP events on the launch stream)",
                    "secondary": []}
            # What actually bounds the kernel: (a) the ISSUE model -- the kernel's own instruction counts from the PMC
            # passes in profiles/ (per launch; valid while the kernel sources hash to what they were collected on) priced at 8 cycles per
            # transcendental, 4 per other vector instruction, 8 of issue hold per MFMA, over 1024 SIMDs at the clock this run held; (b) the
            # MATRIX-ONLY floor at the board's power limit -- the MFMA count at the rate a bare MFMA loop sustains on this board.
            ic = issue_tab.get(f"mha_flash_fwd_L{L_full}_B{2 * B}" + ("" if a.contract == "f32" else "_bf16x3"))
            roof.update({"issue_model_ms": None, "frac_vs_issue_model": None, "matrix_only_floor_ms_at_power_limit": None,
                         "frac_vs_matrix_only_floor": None})      # filled when profiles/ holds instruction counts of THIS shape on THESE kernel sources
            if ic:
                mhz = (clock.summary().get("sclk_mhz_mean") or SCLK_MAX_MHZ)
                valu, mfma, trans = ic["SQ_INSTS_VALU"], ic["SQ_INSTS_MFMA"], ic["SQ_INSTS_VALU_TRANS_F32"]
                cyc = (8.0 * trans + 4.0 * (valu - mfma - trans) + 8.0 * mfma) / N_SIMD
                issue_ms = cyc / (mhz * 1e3)
                roof.update({"issue_model_ms": round(issue_ms, 2), "frac_vs_issue_model": round(issue_ms / avg, 4),
                             "issue_model": {"formula": "(8 SQ_INSTS_VALU_TRANS_F32 + 4 (SQ_INSTS_VALU - SQ_INSTS_MFMA - TRANS) + 8 SQ_INSTS_MFMA) / 1024 SIMDs / sclk",
                                             "sclk_mhz_used": mhz, "counts_per_launch": {k: ic[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32")},
                                             "mfma_coexec_over_busy": (round(ic["SQ_VALU_MFMA_COEXEC_CYCLES"] / ic["SQ_VALU_MFMA_BUSY_CYCLES"], 3)
                                                                       if ic.get("SQ_VALU_MFMA_COEXEC_CYCLES") and ic.get("SQ_VALU_MFMA_BUSY_CYCLES") else None),
                                             "counts_from": ic.get("from", "profiles/roofline_traffic.json _issue_counts")}})
                if a.contract != "f32":
                    floor_ms = mfma / N_SIMD * MFMA16_NS_AT_POWER_LIMIT * 1e-6
                    roof.update({"matrix_only_floor_ms_at_power_limit": round(floor_ms, 2), "frac_vs_matrix_only_floor": round(floor_ms / avg, 4),
                                 "bound_note": "energy: the kernel holds the board at its power limit (device_clock.board_power_w_mean) and the engine clock "
                                               "below its 2400 MHz ceiling; schedules of the same instructions take the same wall time "
                                               "(profiles/r06_attention_bwd_pipeline.txt)"})
            conv_ms = durations.get("conv3x3") or []
            if conv_ms:
                avg = sum(conv_ms) / len(conv_ms)
                fl = 2.0 * 9 * Cc * Cc * L_full * (2 * B)
                ach = fl / (avg * 1e-3) / 1e12
                # the timed launches are the 3x3 convs behind GroupNorm + Swish: in the bf16x3 mode they run the fp16-pair form of
                # the split-operand kernel, THREE 16-bit products per fp32 product (round 3's bf16 triples executed six)
                cpeak = PEAK_F32_MFMA_TFLOPS if a.contract == "f32" else PEAK_BF16_MFMA_TFLOPS / 3
                ckern = ("conv_igemm_kernel (fp32-input MFMA)" if a.contract == "f32" else
                         "conv3x3_x3_kernel<9, false, PAIR> (both operands as fp16 pairs, three products per fp32 product on the "
                         "fp16 MFMA: peak = 16-bit dense MFMA peak / 3)")
                roof["secondary"].append({
                    "bound": "mfma", "kernel": f"hdiff_conv2d_fwd = {ckern} 3x3 {Cc}->{Cc} at {S}x{S}, batch {2 * B} "
                                               "(GroupNorm-Swish prologue, bias/vector/residual epilogue)",
                    "achieved": round(ach, 2), "peak": round(cpeak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / cpeak, 4),
                    "frac_vs_peak_div6": None if a.contract == "f32" else round(ach / (PEAK_BF16_MFMA_TFLOPS / 6), 4),
                    "avg_launch_ms": round(avg, 3),
                    "launches_timed": len(conv_ms), "algorithmic_flop_per_launch": fl,
                    "traffic": traffic_tab.get(f"conv3x3_{Cc}_{S}_B{2 * B}" + ("" if a.contract == "f32" else "_pairs"))})
            gn_ms = durations.get("gn_stats") or []
            if gn_ms:
                avg = sum(gn_ms) / len(gn_ms)
                by = 4.0 * Cc * L_full * (2 * B)                       # the activation is read once; 2*C floats written
                ach = by / (avg * 1e-3) / 1e9
                roof["secondary"].append({
                    "bound": "hbm", "kernel": f"hdiff_gn_scale_shift = gn_stats_kernel + the small merge kernel, {Cc} channels at {S}x{S}, batch {2 * B} "
                                              "(inside the step its input was just written by the producing conv: partly "
                                              "L2 / Infinity-Cache resident; cold-HBM rate: profiles/, tools/gn_once.py)",
                    "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                    "avg_launch_ms": round(avg, 4), "launches_timed": len(gn_ms), "algorithmic_bytes_per_launch": by,
                    "traffic": traffic_tab.get(f"gn_stats_{Cc}_{S}_B{2 * B}")})
        step_tflop = 2 * B * FWD_GFLOP.get(S, 0.0) / 1e3
        out = {
            "metric": "denoising-steps/sec (256x256, T=1000)" if S == 256 else f"denoising-steps/sec ({S}x{S}, T=1000)",
            "value": world * K / elapsed, "unit": "denoising-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.contract == "f32" else "f32 (tensors and accumulation fp32; every operand of the attention contractions as 2xfp16 pieces -- scores: "
                     "a balance per product term, four terms at d_head 16 / three at d_head 32; P.V: three terms -- and of the 3x3 convs behind "
                     "GroupNorm likewise (other 3x3 convs: 3xbf16 pieces), on the 16-bit MFMA; "
                     "fp32-class error: golden suite green at the fp32 tolerances, per-kernel error vs float64 <= 1.25x (attention) / "
                     "1.5x (conv) the fp32-MFMA kernel's)",
            "data": "synthetic",
            "config": {"workload": f"CFG-DDPM sampling, {S}x{S}, T=1000 linear schedule (1e-4..0.02), w={GUIDANCE_W}, "
                                   f"batch {B}/GPU (2x{B} UNet forwards per step; BASELINE's metric names no batch: 8 = the per-GPU batch of its config C5), default UNet ch=128 ch_mult=[1,2,2,2] "
                                   "num_res_blocks=2 (47.8 M params), random-init weights, in-kernel Philox noise",
                       "batch_per_gpu": B, "image": S, "launch": "plain launches" if a.eager else "hipGraph replay of the captured step",
                       "ms_per_step_plain_launches_with_events": None if kernel_pass_s is None else kernel_pass_s * 1e3,
                       "sample_steps_per_s": world * K * B / elapsed,
                       "algorithmic_tflop_per_step_per_gpu": step_tflop,
                       "whole_step_tflops_per_gpu": step_tflop / (elapsed / K) if elapsed > 0 else None,
                       "attention_contract": a.contract, "other_contract_mode": alt},
            "roofline": roof,
            "device_clock": clock.summary(),
            "box_range": BOX_RANGE,
        }
    # everything below is outside the timed region and never enters `value`; N = 1 only
    if rank == 0 and world == 1:
        del plan, sp, sampler, model, x_timed
        _release()
        if not a.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(S, B, dev)
            _release()
        if mode_parity is not None:
            out.setdefault("parity", {})["modes"] = mode_parity
        if not a.no_extras:
            try:
                out["configs"] = other_configs(dev, small=a.extras_scale == "small")
                c1 = out["configs"].get("C1")
                if c1 and "cpu_baseline" in out:
                    out["cpu_baseline"]["same_config"] = {k: c1[k] for k in ("workload", "gpu_steps_per_s", "cpu_steps_per_s",
                                                                             "cpu_s", "cpu_cores", "cpu_kind", "gpu_over_cpu")}
                if c1 and "parity" in out:
                    out["parity"]["end_to_end_C1"] = c1["parity"]
            except Exception as e:     # the headline line must survive a failure of an auxiliary leg -- but say so loudly
                import traceback
                traceback.print_exc()
                out["configs"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
