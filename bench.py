#!/usr/bin/env python3
"""Headline benchmark: denoising-steps/sec of the CFG-DDPM sampler at 256x256 on the T=1000 schedule.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one iteration of the reference sampler loop (DiffusionCondition.py:87-96) for the whole per-GPU batch:
cond + uncond UNet forward (one 2B-batched launch sequence) + fused CFG/posterior update.  Inputs (x_T, labels, weights)
are resident in HBM before the timed region.  Sampling shards by image: every rank runs its own batch with its own seed
and there is no data-path collective (weak scaling); value = N * K / max-over-ranks wall time.

The timed region replays the hipGraph-captured step (the north-star formulation; ``--eager`` times plain launches instead).
Per-kernel durations come from a second pass of the same K steps issued as plain launches with HIP events recorded on the
launch stream around the launches of interest (events cannot bracket nodes inside a graph replay; the kernels, their
arguments and their order are identical -- the two passes agree to < 0.1 % at this size, both wall times are reported).

Rank 0 prints ONE JSON line with the contract keys plus
  roofline      dominant kernel (flash attention at L = 65536, d_head = 16) against the fp32-MFMA peak; ``secondary`` holds
                the full-resolution 3x3 convolution (MFMA) and the GroupNorm statistics pass (HBM) the same way
  cpu_baseline  the CPU oracle (oracle/cpu_path.py, kind "port"): one whole UNet forward at 128x128 timed on the host
                cores (a bounded sample, ~25 s), scaled to the benchmark's step by the algorithmic FLOP ratio
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

MODEL = dict(T=1000, num_labels=10, ch=128, ch_mult=[1, 2, 2, 2], num_res_blocks=2, dropout=0.15)
BETA = (1e-4, 0.02)
GUIDANCE_W = 1.8
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* dense peak (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2516.6    # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s is what a float4 copy achieves
# algorithmic FLOPs per sample per UNet forward (torch flop counter on the reference UNet)
FWD_GFLOP = {64: 74.0, 128: 529.6, 256: 5857.4, 512: 83254.1}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contract", choices=["f32", "bf16x3"], default="f32",
                    help="attention contractions: fp32-input MFMA (default) or the fp32-class three-piece bf16 split")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra (untimed in `value`) run in the other contraction mode")
    ap.add_argument("--eager", action="store_true", help="time plain launches instead of the hipGraph replay")
    ap.add_argument("--graph", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the second, event-instrumented pass (no roofline)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="dev: run the N > 1 code path (rendezvous, barriers, max over ranks, aggregation) with every rank on "
                         "cuda:0 over gloo -- RCCL refuses two ranks on one device; the numbers mean nothing")
    return ap.parse_args()


def cpu_baseline(size, batch):
    """Oracle on the host cores: ONE whole UNet forward at 128x128, B = 1 (the largest size that finishes in tens of
    seconds; attention is 60 % of its FLOPs against 85 % at 256x256), scaled to one denoising step of the benchmark
    (2 * batch forwards at `size`) by the algorithmic FLOP ratio."""
    from oracle import cpu_path as O
    threads = min(os.cpu_count() or 1, 16)          # the one-GPU box's CPU share
    torch.set_num_threads(threads)
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    torch.manual_seed(0)
    m = UNet(**MODEL).eval()
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    cfg = O.UNetConfig(T=MODEL["T"], num_labels=MODEL["num_labels"], ch=MODEL["ch"], ch_mult=tuple(MODEL["ch_mult"]),
                       num_res_blocks=MODEL["num_res_blocks"])
    S = 128
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, S, S, generator=g)
    with torch.no_grad():
        O.unet_forward(sd, cfg, torch.randn(1, 3, 32, 32, generator=g), torch.tensor([5]), torch.tensor([1]))  # warm
        t0 = time.perf_counter()
        O.unet_forward(sd, cfg, x, torch.tensor([500]), torch.tensor([1]))
        dt = time.perf_counter() - t0
    fwd = FWD_GFLOP[size] if size in FWD_GFLOP else FWD_GFLOP[S] * (size / S) ** 4
    scale = 2 * batch * fwd / FWD_GFLOP[S]
    return {"value": 1.0 / (dt * scale), "unit": "denoising-steps/s", "cores": threads, "kind": "port",
            "sample": f"one UNet forward at {S}x{S}, B=1 on the CPU oracle (blockwise attention) took {dt:.1f} s = "
                      f"{FWD_GFLOP[S] / dt:.0f} GFLOP/s; one step of the benchmark is 2x{batch} forwards at {size}x{size} = "
                      f"{scale:.0f}x its FLOPs (the reference's own materialised-attention formulation cannot run at "
                      "256x256: 137 GB/sample)",
            "measured_s": dt, "torch_threads": threads}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    from hdiff_amd.DiffusionFreeGuidence.DiffusionCondition import GaussianDiffusionSampler, _SamplerPlan
    lib = hdiff_amd.lib()

    S, B, K, Wm = a.size, a.batch, a.steps, a.warmup
    hdiff_amd.set_contraction_mode(a.contract)
    torch.manual_seed(0)                              # same weights on every rank (replicated model)
    model = UNet(**MODEL).eval().to(dev)
    sampler = GaussianDiffusionSampler(model, BETA[0], BETA[1], MODEL["T"], w=GUIDANCE_W).to(dev)
    g = torch.Generator().manual_seed(1234 + rank)    # per-rank data
    x_T = torch.randn(B, 3, S, S, generator=g).to(dev)
    labels = (torch.arange(B) % 2 + 1).to(dev)

    with torch.no_grad():
        sp = _SamplerPlan(sampler, B, S, S, dev)
        plan = sp.variant(False, 1234 + rank)
        sp.x.copy_(x_T)
        sp.unet.labels.copy_(torch.cat([labels, torch.zeros_like(labels)]))
        sp.step.fill_(MODEL["T"] - 1)
        sp.nan_flag.zero_()
        stream = torch.cuda.current_stream(dev).cuda_stream
        use_graph = not a.eager
        if use_graph:
            plan.capture()

        # launches of interest: attention over the full-resolution token set (dominant), the full-resolution 128 -> 128 3x3
        # convolutions, the full-resolution GroupNorm statistics passes
        L_full = S * S
        Cc = MODEL["ch"] * MODEL["ch_mult"][0]
        groups = {"attn": [], "conv3x3": [], "gn_stats": []}
        for i, (name, _, args) in enumerate(plan.ops):
            if name == "hdiff_mha_flash_fwd" and args[6] == L_full:
                groups["attn"].append(i)
            elif name == "hdiff_conv2d_fwd":
                d = args[0]._obj
                if d.ntaps == 9 and d.C0 == Cc and d.C1 == 0 and d.Cout == Cc and d.H == S and d.in_stride == 1:
                    groups["conv3x3"].append(i)
            elif name == "hdiff_gn_stats" and args[5] == L_full and args[2] + args[3] == Cc:
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

        for _ in range(Wm):
            one_step()
        if dist:
            td.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(K):
            one_step()
        torch.cuda.synchronize(dev)
        if dist:
            td.barrier()
        elapsed = time.perf_counter() - t0
        assert int(sp.nan_flag.item()) == 0, "nan in tensor."
        assert int(sp.step.item()) == MODEL["T"] - 1 - (K + Wm)

        # second pass: the same K steps as plain launches with HIP events around the launches of interest
        kernel_pass_s = None
        if not a.no_kernel_pass:
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(K):
                one_step(instrumented=True)
            torch.cuda.synchronize(dev)
            kernel_pass_s = (time.perf_counter() - t1) / K
            assert int(sp.nan_flag.item()) == 0, "nan in tensor."

        # the other contraction mode, reported beside the headline (never part of `value`)
        alt = None
        if world == 1 and not a.no_alt:
            other = "bf16x3" if a.contract == "f32" else "f32"
            hdiff_amd.set_contraction_mode(other)
            use_graph = False                         # a captured graph bakes the contraction mode: plain launches here
            one_step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(K):
                one_step()
            torch.cuda.synchronize(dev)
            dt_alt = (time.perf_counter() - t1) / K
            hdiff_amd.set_contraction_mode(a.contract)
            assert int(sp.nan_flag.item()) == 0, "nan in tensor."
            alt = {"contract": other, "ms_per_step": dt_alt * 1e3, "denoising_steps_per_s": 1.0 / dt_alt,
                   "note": "same workload with the attention contractions in the other mode; bf16x3 = every fp32 operand as "
                           "three bf16 pieces, six products on the bf16 MFMA, fp32 accumulate (fp32-class error, "
                           "tests/test_gpu_ops.py::test_flash_attention_split_bf16_is_fp32_class)"}

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
        traffic_tab = {}
        prof = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.isfile(prof):
            try:
                traffic_tab = json.load(open(prof))
            except Exception:
                traffic_tab = {}
        flops_per_launch = 4.0 * L_full * L_full * Cc * (2 * B)          # QK^T + PV over 8 heads, 2B samples (CFG)
        roof = None
        att_ms = durations.get("attn") or []
        if att_ms:
            avg = sum(att_ms) / len(att_ms)
            ach = flops_per_launch / (avg * 1e-3) / 1e12
            if a.contract == "f32":
                kname, peak = "mha_flash_fwd_fast_kernel<16,4>", PEAK_F32_MFMA_TFLOPS
            else:   # six bf16 products per fp32 product: the scheme's fp32-equivalent ceiling is the bf16 dense peak / 6
                kname, peak = "mha_flash_fwd_x3_kernel<16,4> (3xbf16 split, peak = bf16 dense peak / 6)", PEAK_BF16_MFMA_TFLOPS / 6
            roof = {"bound": "mfma",
                    "kernel": f"hdiff_mha_flash_fwd = {kname} + overflow-check pass, L={L_full} d_head=16 "
                              f"heads=8 batch={2 * B}",
                    "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic_tab.get(f"mha_flash_fwd_L{L_full}_B{2 * B}"),
                    "avg_launch_ms": round(avg, 3), "launches_timed": len(att_ms),
                    "algorithmic_flop_per_launch": flops_per_launch,
                    "timed_in": "second pass of the same K steps as plain launches (HIP events on the launch stream)",
                    "secondary": []}
            conv_ms = durations.get("conv3x3") or []
            if conv_ms and a.contract == "f32":
                avg = sum(conv_ms) / len(conv_ms)
                fl = 2.0 * 9 * Cc * Cc * L_full * (2 * B)
                ach = fl / (avg * 1e-3) / 1e12
                roof["secondary"].append({
                    "bound": "mfma", "kernel": f"hdiff_conv2d_fwd = conv_igemm_kernel 3x3 {Cc}->{Cc} at {S}x{S}, batch {2 * B} "
                                               "(GroupNorm-Swish prologue, bias/vector/residual epilogue)",
                    "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "avg_launch_ms": round(avg, 3),
                    "launches_timed": len(conv_ms), "algorithmic_flop_per_launch": fl,
                    "traffic": traffic_tab.get(f"conv3x3_{Cc}_{S}_B{2 * B}")})
            gn_ms = durations.get("gn_stats") or []
            if gn_ms:
                avg = sum(gn_ms) / len(gn_ms)
                by = 4.0 * Cc * L_full * (2 * B)                       # the activation is read once; 2*C floats written
                ach = by / (avg * 1e-3) / 1e9
                roof["secondary"].append({
                    "bound": "hbm", "kernel": f"hdiff_gn_stats = gn_stats_kernel, {Cc} channels at {S}x{S}, batch {2 * B} "
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
            "dtype": "f32" if a.contract == "f32" else "f32 (attention products as 3xbf16 split, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"CFG-DDPM sampling, {S}x{S}, T=1000 linear schedule (1e-4..0.02), w={GUIDANCE_W}, "
                                   f"batch {B}/GPU (2x{B} UNet forwards per step), default UNet ch=128 ch_mult=[1,2,2,2] "
                                   "num_res_blocks=2 (47.8 M params), random-init weights, in-kernel Philox noise",
                       "batch_per_gpu": B, "image": S, "launch": "plain launches" if a.eager else "hipGraph replay of the captured step",
                       "ms_per_step_plain_launches_with_events": None if kernel_pass_s is None else kernel_pass_s * 1e3,
                       "sample_steps_per_s": world * K * B / elapsed,
                       "algorithmic_tflop_per_step_per_gpu": step_tflop,
                       "whole_step_tflops_per_gpu": step_tflop / (elapsed / K) if elapsed > 0 else None,
                       "attention_contract": a.contract, "other_contract_mode": alt},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, B)
        print(json.dumps(out), flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
