"""Drop-in for the reference's ``DiffusionFreeGuidence/DiffusionCondition.py``: ``extract``, ``GaussianDiffusionTrainer``
(Algorithm 1) and ``GaussianDiffusionSampler`` (Algorithm 2, classifier-free guidance) with the same constructor /
``forward`` signatures, registered float64 buffers and error behaviour.

What differs underneath (reference lines in brackets):
  * the cond + uncond denoiser calls of one step [:76-77] run as ONE 2B-batched UNet plan (labels = [labels; 0]);
  * CFG combine, posterior mean, noise add and the NaN check [:78-79, :91-96] are one kernel (``hdiff_ddpm_step``);
  * one whole denoising step is captured into a hipGraph and replayed T times with a device-resident step counter
    [:87-96]; the per-step ``print`` [:88] is dropped and the per-step NaN ``assert`` [:96] is evaluated once after the
    loop from a device flag (same ``AssertionError("nan in tensor.")``);
  * per-step noise is drawn in-kernel (Philox4x32-10 + Box-Muller) from a seed taken from torch's default generator,
    so ``torch.manual_seed`` still makes sampling reproducible.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
import warnings

from .. import _capi
from .. import engine as E


__all__ = ["extract", "GaussianDiffusionTrainer", "GaussianDiffusionSampler"]


def extract(v, t, x_shape):
    """Coefficients at the given timesteps, cast float64 -> fp32 AFTER the gather, shaped [B,1,1,...] (reference :9-16)."""
    out = torch.gather(v, index=t, dim=0).float().to(t.device)
    return out.view([t.shape[0]] + [1] * (len(x_shape) - 1))


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _gpu_input(x, name):
    """A caller's tensor as the kernels need it: on the GPU (no CPU fallback), fp32 / integer, contiguous NCHW."""
    if x.is_cuda and not x.is_contiguous():
        x = x.contiguous()              # the reference accepts strided views; the kernels read dense NCHW
    E.require_gpu_tensor(x, name)
    return x


def _timesteps(t, T: int, device):
    """The index vector of ``extract`` (reference :9-16) as the kernels read it: int64, contiguous, on ``device``, and in
    range -- ``torch.gather`` raises for an index outside [0, T), so does this (the kernels clamp on top: a bad index can
    never fault the GPU)."""
    if not torch.is_tensor(t) or t.dtype not in (torch.int64, torch.int32):
        raise RuntimeError("gather(): Expected dtype int64 for index")
    t = t.to(device=device, dtype=torch.int64).contiguous()
    if t.numel():
        lo, hi = torch.stack([t.min(), t.max()]).tolist()
        if lo < 0 or hi >= T:
            raise RuntimeError(f"index {hi if hi >= T else lo} is out of bounds for dimension 0 with size {T}")
    return t


class GaussianDiffusionTrainer(nn.Module):
    """forward(x_0, labels) -> unreduced squared error of the eps prediction (reference :19-46)."""

    def __init__(self, model, beta_1, beta_T, T):
        super().__init__()
        self.model = model
        self.T = T
        self.register_buffer('betas', torch.linspace(beta_1, beta_T, T).double())
        alphas_bar = torch.cumprod(1. - self.betas, dim=0)
        self.register_buffer('sqrt_alphas_bar', torch.sqrt(alphas_bar))
        self.register_buffer('sqrt_one_minus_alphas_bar', torch.sqrt(1. - alphas_bar))

    def forward(self, x_0, labels, *, t=None, noise=None):
        """``t`` / ``noise`` may be injected (parity tests); by default they are drawn exactly where the reference
        draws them (``torch.randint`` then ``torch.randn_like``, reference :41-42)."""
        x_0, labels = _gpu_input(x_0, "x_0"), _gpu_input(labels, "labels")
        lib = _capi.lib()
        B = int(x_0.shape[0])
        if t is None:
            t = torch.randint(self.T, size=(B,), device=x_0.device)     # in range by construction: no device->host read
        else:
            t = _timesteps(t, self.T, x_0.device)                        # a caller's vector is validated like torch.gather would
        if noise is None:
            noise = torch.randn_like(x_0)
        noise = _gpu_input(noise, "noise")
        assert noise.shape == x_0.shape
        sa, sb = self.sqrt_alphas_bar.float(), self.sqrt_one_minus_alphas_bar.float()
        x_t = torch.empty_like(x_0)
        with torch.cuda.device(x_0.device):
            s = _stream(x_0.device)
            _capi.check(lib.hdiff_q_sample(x_0.data_ptr(), noise.data_ptr(), t.data_ptr(), sa.data_ptr(), sb.data_ptr(),
                                           x_t.data_ptr(), B, x_0.numel() // B, self.T, s), "q_sample")
            eps_hat = self.model(x_t, t, labels)
            if eps_hat.requires_grad:
                from ..autograd import sq_err_with_grad
                return sq_err_with_grad(eps_hat, noise)
            loss = torch.empty_like(x_0)
            _capi.check(lib.hdiff_sq_err(eps_hat.data_ptr(), noise.data_ptr(), loss.data_ptr(), x_0.numel(), s), "sq_err")
            return loss


class _SamplerPlan:
    """One captured denoising step for a fixed (B, H, W): 2B UNet -> fused DDPM update (which also prepares the next step)."""

    def __init__(self, sampler: "GaussianDiffusionSampler", B: int, H: int, W: int, device):
        model = sampler.model
        self.unet = model.plan_for(2 * B, H, W, device)
        up = self.unet
        n = B * 3 * H * W
        dev = device
        self.x = torch.empty(B, 3, H, W, device=dev)
        self.noise = torch.empty(B, 3, H, W, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.nan_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.done = torch.zeros(1, dtype=torch.int32, device=dev)       # finished-workgroup counter of the fused update
        var = torch.cat([sampler.posterior_var[1:2], sampler.betas[1:]])                 # reference :74
        self.c1 = sampler.coeff1.float().contiguous()                                     # extract(): f64 -> f32
        self.c2 = sampler.coeff2.float().contiguous()
        self.sigma = torch.sqrt(var.float()).contiguous()                                 # sqrt on the fp32 value (:95)
        self.seed = 0
        self.B, self.n = B, n
        self._variants = {}
        self._sampler = sampler

    def _build(self, inject_noise: bool) -> E.Plan:
        """One denoising step = the 2B UNet launches + ONE fused update kernel.  The update also writes x_next into both
        halves of the UNet's input, decrements the device-resident step and refills the time vector (hdiff_ddpm_step_loop),
        so nothing else has to run between two replays; `reset()` puts the loop state at its start."""
        up, B, n = self.unet, self.B, self.n
        p = E.Plan(self.x.device)
        p.ops.extend(up.plan.ops)
        eps = up.out
        d = _capi.DdpmLoopDesc()
        d.x, d.eps_c, d.eps_u = self.x.data_ptr(), eps.data_ptr(), eps.data_ptr() + 4 * n
        d.noise = self.noise.data_ptr() if inject_noise else None
        d.x_next = self.x.data_ptr()
        d.coeff1, d.coeff2, d.sigma = self.c1.data_ptr(), self.c2.data_ptr(), self.sigma.data_ptr()
        d.step_ptr, d.T = self.step.data_ptr(), int(self._sampler.T)
        d.w, d.seed = float(self._sampler.w), self.seed
        d.nan_flag, d.n = self.nan_flag.data_ptr(), n
        d.x_dup0, d.x_dup1 = up.x.data_ptr(), up.x.data_ptr() + 4 * n
        d.t_next, d.t_count = up.t.data_ptr(), 2 * B
        d.done_counter = self.done.data_ptr()
        p.keep(d)
        p.call("hdiff_ddpm_step_loop", C.byref(d))
        return p

    def reset(self, x_T: torch.Tensor, labels: torch.Tensor, step: Optional[int] = None) -> None:
        """Loop state at its start: x = x_T (also in both halves of the UNet input), labels = [labels; 0] (reference :76-77),
        the time vector and the device-resident step at T - 1 (or `step`), flags cleared."""
        T = int(self._sampler.T)
        step = T - 1 if step is None else int(step)
        self.x.copy_(x_T)
        self.unet.x[:self.B].copy_(x_T)
        self.unet.x[self.B:].copy_(x_T)
        self.unet.labels.copy_(torch.cat([labels, torch.zeros_like(labels)], dim=0))
        self.unet.t.fill_(step)
        self.step.fill_(step)
        self.nan_flag.zero_()
        self.done.zero_()

    def variant(self, inject_noise: bool, seed: int) -> E.Plan:
        # the guidance weight is a launch argument of the fused update: the reference reads self.w on every step (:78), so
        # a changed sampler.w must rebuild the captured step
        key = (inject_noise, seed if not inject_noise else 0, _capi.lib().hdiff_get_contraction_mode(),
               float(self._sampler.w))
        if key not in self._variants:
            self.seed = seed
            self._variants.clear()          # a graph bakes its seed (and the contraction mode): keep one live variant
            self._variants[key] = self._build(inject_noise)
        return self._variants[key]


class GaussianDiffusionSampler(nn.Module):
    """forward(x_T, labels) -> x_0 clipped to [-1, 1] by T ancestral steps with classifier-free guidance (reference :49-98)."""

    def __init__(self, model, beta_1, beta_T, T, w=0.):
        super().__init__()
        self.model = model
        self.T = T
        self.w = w
        self.register_buffer('betas', torch.linspace(beta_1, beta_T, T).double())
        alphas = 1. - self.betas
        alphas_bar = torch.cumprod(alphas, dim=0)
        alphas_bar_prev = F.pad(alphas_bar, [1, 0], value=1)[:T]
        self.register_buffer('coeff1', torch.sqrt(1. / alphas))
        self.register_buffer('coeff2', self.coeff1 * (1. - alphas) / torch.sqrt(1. - alphas_bar))
        self.register_buffer('posterior_var', self.betas * (1. - alphas_bar_prev) / (1. - alphas_bar))
        self.use_graph = True
        self._splans = {}

    # -- single-step API of the reference -----------------------------------------------------------------------------
    def predict_xt_prev_mean_from_eps(self, x_t, t, eps):
        assert x_t.shape == eps.shape
        x_t, eps = _gpu_input(x_t, "x_t"), _gpu_input(eps, "eps")
        t = _timesteps(t, self.T, x_t.device)
        lib = _capi.lib()
        B = int(x_t.shape[0])
        c1 = self.coeff1.float()
        neg_c2 = -(self.coeff2.float())
        out = torch.empty_like(x_t)
        # coeff1[t]*x_t - coeff2[t]*eps  ==  coeff1[t]*x_t + (-coeff2[t])*eps with identical roundings
        with torch.cuda.device(x_t.device):
            _capi.check(lib.hdiff_q_sample(x_t.data_ptr(), eps.data_ptr(), t.data_ptr(), c1.data_ptr(), neg_c2.data_ptr(),
                                           out.data_ptr(), B, x_t.numel() // B, self.T, _stream(x_t.device)),
                        "posterior_mean")
        return out

    def _paired_eps(self, x_t, t, labels):
        """cond and uncond denoiser outputs from one 2B-batched launch sequence (reference :76-77)."""
        x2 = torch.cat([x_t, x_t], dim=0)
        t2 = torch.cat([t, t], dim=0)
        l2 = torch.cat([labels, torch.zeros_like(labels)], dim=0)
        e2 = self.model(x2, t2, l2)
        B = x_t.shape[0]
        return e2[:B], e2[B:]

    def p_mean_variance(self, x_t, t, labels):
        x_t, labels = _gpu_input(x_t, "x_t"), _gpu_input(labels, "labels")
        t = _timesteps(t, self.T, x_t.device)
        var = torch.cat([self.posterior_var[1:2], self.betas[1:]])
        var = extract(var, t, x_t.shape)
        eps_c, eps_u = self._paired_eps(x_t, t, labels)
        lib = _capi.lib()
        eps = torch.empty_like(x_t)
        with torch.cuda.device(x_t.device):
            _capi.check(lib.hdiff_axpby(C.c_float(1. + self.w), eps_c.contiguous().data_ptr(), C.c_float(-self.w),
                                        eps_u.contiguous().data_ptr(), eps.data_ptr(), x_t.numel(), _stream(x_t.device)),
                        "cfg_combine")
        return self.predict_xt_prev_mean_from_eps(x_t, t, eps=eps), var

    # -- the loop -----------------------------------------------------------------------------------------------------
    def forward(self, x_T, labels, *, noise_by_step=None, trajectory: Optional[List[torch.Tensor]] = None):
        """``noise_by_step[time_step]`` injects the per-step z (parity tests); ``trajectory`` collects the pre-clip
        x_t after every step.  Both default to the reference behaviour."""
        x_T, labels = _gpu_input(x_T, "x_T"), _gpu_input(labels, "labels")
        if torch.is_grad_enabled():
            # The reference runs here too (DiffusionCondition.py:82-98) and records an autograd graph through all 2T model
            # evaluations, which nothing in its callers ever differentiates (TrainCondition.eval samples under no_grad).
            # The loop here is an inference loop: it runs without a graph and hands back a detached tensor.  A caller that
            # asks for a gradient with respect to x_T clearly expects that graph: refused; enabled autograd with trainable
            # parameters alone: said once per sampler instance.
            if x_T.requires_grad:
                raise RuntimeError("GaussianDiffusionSampler.forward: x_T requires grad, but the denoising loop runs under "
                                   "torch.no_grad() and cannot be differentiated (the reference would record a graph through all "
                                   "2T model evaluations); detach x_T or call under torch.no_grad()")
            if any(p.requires_grad for p in self.model.parameters()) and not getattr(self, "_warned_grad", False):
                self._warned_grad = True
                warnings.warn("GaussianDiffusionSampler.forward was called with autograd enabled: the denoising loop runs under "
                              "torch.no_grad() and returns a tensor without grad_fn (the reference would record a graph "
                              "through all 2T model evaluations)", RuntimeWarning, stacklevel=2)
        with torch.no_grad(), torch.cuda.device(x_T.device):
            return self._forward(x_T, labels, noise_by_step, trajectory)

    def _forward(self, x_T, labels, noise_by_step, trajectory):
        lib = _capi.lib()
        B, Cx, H, W = (int(v) for v in x_T.shape)
        dev = x_T.device
        key = (B, H, W, str(dev))
        sp = self._splans.get(key)
        if sp is None or sp.unet is not self.model.plan_for(2 * B, H, W, dev):
            sp = _SamplerPlan(self, B, H, W, dev)
            self._splans = {key: sp}
        # One pack per call is nothing against T steps, and it closes the one hole of version-keyed staleness: a write through
        # ``p.data`` (EMA swaps, ``dist.broadcast(p.data)``) does not bump ``p._version``.
        sp.unet.plan.pack_weights()
        self.model.check_indices(torch.zeros_like(labels), labels)
        inject = noise_by_step is not None
        seed = 0 if inject else int(torch.empty((), dtype=torch.int64).random_().item())
        plan = sp.variant(inject, seed)
        sp.reset(x_T, labels)
        graphed = self.use_graph and trajectory is None
        if graphed:
            plan.capture()
        for time_step in reversed(range(self.T)):
            if inject and time_step > 0:
                sp.noise.copy_(noise_by_step[time_step])
            if graphed:
                plan.replay()
            else:
                plan.run()
            if trajectory is not None:
                trajectory.append(sp.x.clone())
        assert int(sp.nan_flag.item()) == 0, "nan in tensor."
        out = torch.empty_like(x_T)
        _capi.check(lib.hdiff_clip(sp.x.data_ptr(), out.data_ptr(), C.c_float(-1.0), C.c_float(1.0), x_T.numel(),
                                   _stream(dev)), "clip")
        return out
