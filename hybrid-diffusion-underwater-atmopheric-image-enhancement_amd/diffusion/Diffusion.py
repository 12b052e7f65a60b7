"""Drop-in for the sampler of the reference's ``diffusion/Diffusion.py``: ``extract`` (:16-23) and
``GaussianDiffusionSampler`` (:182-269) -- the image-conditioned ancestral sampler and the deterministic DDIM sampler --
with the same constructor / ``forward`` signature, registered float64 buffers and plain attributes.

Underneath: one DynamicUNet evaluation + one fused update kernel per step, the whole step captured into a hipGraph and
replayed (device-resident step counter, time-step and coefficient tables); y_t is updated in place in the plan's buffer.

Kept as written in the reference, on purpose:
  * the model is called as ``model(input, t)``: no label, ``context_zero=True``;
  * with ``unconditional_guidance_scale != 1`` the reference evaluates that same function twice and forms
    ``eps_u + s * (eps - eps_u)`` (:254-257); both evaluations being identical, that is ``eps_u + s * 0 = eps`` exactly,
    so one evaluation is issued here;
  * DDIM lays its time steps over a literal 1000 and reads ``alphas_bar[t + 1]`` (:243-251); ``eta = 0`` makes the
    per-step ``c1 * randn`` term an exact zero (:260-263).
The trainer of this file (:26-180) needs pretrained VGG / DINO / LPIPS networks and is out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _capi
from .. import engine as E

__all__ = ["extract", "GaussianDiffusionSampler"]


def extract(v, t, x_shape):
    """Coefficients at the given timesteps, cast to fp32 AFTER the gather, shaped [B,1,1,...] (reference :16-23)."""
    device = t.device
    out = torch.gather(v, index=t, dim=0).float().to(device)
    return out.view([t.shape[0]] + [1] * (len(x_shape) - 1))


class _StepPlan:
    """One captured sampling step for a fixed (B, H, W) and mode: fill t -> DynamicUNet -> update -> advance the counter."""

    def __init__(self, sampler: "GaussianDiffusionSampler", B: int, H: int, W: int, device, ddim_step: Optional[int],
                 inject_noise: bool = False, seed: int = 0):
        self.unet = sampler.model.plan_for(B, H, W, device, True)
        up = self.unet
        if up.out_hw != (H, W):
            raise RuntimeError(f"The size of tensor a ({W}) must match the size of tensor b ({up.out_hw[1]}) at non-singleton "
                               "dimension 3")
        n = B * 3 * H * W
        self.B, self.n = B, n
        self.step = torch.zeros(1, dtype=torch.int32, device=device)
        self.nan_flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.noise = torch.empty(B, 3, H, W, device=device)
        p = E.Plan(device)
        if ddim_step is None:
            # ancestral (:224-236): t = the step counter itself; the first tree's fused update with w = 0 (eps_u := eps)
            var = torch.cat([sampler.posterior_var[1:2], sampler.betas[1:]])                      # :210
            self.c1 = sampler.coeff1.float().contiguous()
            self.c2 = sampler.coeff2.float().contiguous()
            self.sigma = torch.sqrt(var.float()).contiguous()
            self.n_steps = sampler.T
            p.call("hdiff_fill_t", up.t.data_ptr(), self.step.data_ptr(), B)
            p.ops.extend(up.plan.ops)
            # noise: injected buffer (parity runs) or drawn in-kernel (Philox, counter = (seed, step, element))
            p.call("hdiff_ddpm_step", up.y.data_ptr(), up.out.data_ptr(), up.out.data_ptr(),
                   self.noise.data_ptr() if inject_noise else None, up.y.data_ptr(), self.c1.data_ptr(), self.c2.data_ptr(),
                   self.sigma.data_ptr(), self.step.data_ptr(), int(sampler.T), C.c_double(0.0), C.c_uint64(seed),
                   self.nan_flag.data_ptr(), n)
        else:
            step = int(1000 / ddim_step)                                                           # :243-247
            seq = list(range(0, 1000, step))
            seq_next = [-1] + seq[:-1]
            ab = sampler.alphas_bar
            if seq[-1] + 1 >= ab.shape[0]:
                raise RuntimeError(f"index {seq[-1] + 1} is out of bounds for dimension 0 with size {ab.shape[0]}")
            ab = ab.to(device)
            # tables indexed by the down-counting step counter k (k = len-1 first): same fp32 ops as :250-262
            at = ab[torch.tensor(seq, device=device) + 1].float()
            at_next = ab[torch.tensor(seq_next, device=device) + 1].float()
            c1 = 0 * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
            c2 = ((1 - at_next) - c1 ** 2).sqrt()
            self.tab = torch.stack([(1 - at).sqrt(), at.sqrt(), at_next.sqrt(), c2], dim=1).contiguous()
            self.t_tab = torch.tensor(seq, dtype=torch.int32, device=device)
            self.n_steps = len(seq)
            p.call("hdiff_fill_from_table", up.t.data_ptr(), self.t_tab.data_ptr(), self.step.data_ptr(), self.n_steps, B)
            p.ops.extend(up.plan.ops)
            p.call("hdiff_ddim_step", up.y.data_ptr(), up.out.data_ptr(), up.y.data_ptr(), self.tab.data_ptr(),
                   self.step.data_ptr(), self.n_steps, self.nan_flag.data_ptr(), n)
        p.call("hdiff_step_decrement", self.step.data_ptr())
        self.plan = p


class GaussianDiffusionSampler(nn.Module):
    """forward(input_image, ddim=False, unconditional_guidance_scale=1, ddim_step=None) -> enhanced image clipped to [-1, 1]
    (reference diffusion/Diffusion.py:182-269).  ``input_image`` is in [0, 255] (divided by 255 here, as there)."""

    GRAPH_MIN_STEPS = 4

    def __init__(self, model, beta_1, beta_T, T):
        super().__init__()
        self.model = model
        self.T = T
        self.register_buffer('betas', torch.linspace(beta_1, beta_T, T).double())
        alphas = 1. - self.betas
        alphas_bar = torch.cumprod(alphas, dim=0)
        alphas_bar_prev = F.pad(alphas_bar, [1, 0], value=1)[:T]
        self.sqrt_alphas_bar = alphas_bar                      # sic: the reference stores alphas_bar under this name (:193)
        self.sqrt_one_minus_alphas_bar = torch.sqrt(1. - alphas_bar)
        self.alphas_bar = alphas_bar
        self.one_minus_alphas_bar = (1. - alphas_bar)
        self.register_buffer('coeff1', torch.sqrt(1. / alphas))
        self.register_buffer('coeff2', self.coeff1 * (1. - alphas) / torch.sqrt(1. - alphas_bar))
        self.register_buffer('posterior_var', self.betas * (1. - alphas_bar_prev) / (1. - alphas_bar))
        self._plans = {}

    def predict_xt_prev_mean_from_eps(self, t, eps, y_t):
        assert y_t.shape == eps.shape
        E.require_gpu_tensor(y_t, "y_t")
        E.require_gpu_tensor(eps, "eps")
        lib = _capi.lib()
        out = torch.empty_like(y_t)
        c1, c2 = extract(self.coeff1, t, y_t.shape), extract(self.coeff2, t, y_t.shape)
        s = torch.cuda.current_stream(y_t.device).cuda_stream
        per = y_t.numel() // y_t.shape[0]
        for b in range(y_t.shape[0]):      # tiny helper path (the sampler itself uses the fused per-step kernel)
            _capi.check(lib.hdiff_axpby(C.c_float(float(c1[b])), y_t[b].data_ptr(), C.c_float(-float(c2[b])), eps[b].data_ptr(),
                                        out[b].data_ptr(), per, s), "axpby")
        return out

    def p_mean_variance(self, input, t, y_t):
        var = torch.cat([self.posterior_var[1:2], self.betas[1:]])
        var = extract(var, t, input.shape)
        eps = self.model(input, t)
        return self.predict_xt_prev_mean_from_eps(t, eps, y_t), var

    def forward(self, input_image, ddim=False, unconditional_guidance_scale=1, ddim_step=None, *, y_T=None,
                noise_by_step: Optional[List[torch.Tensor]] = None, trajectory: Optional[List[torch.Tensor]] = None):
        """``y_T`` / ``noise_by_step`` inject the random draws (parity runs; ``noise_by_step[k]`` is the k-th per-step draw of
        the ancestral loop, in call order); by default they come from torch's generator exactly where the reference draws
        them.  ``trajectory`` collects the pre-clip y_t after every step.  Called with autograd enabled (the reference would
        record a graph through every model evaluation, diffusion/Diffusion.py:217-269) the loop still runs without one and
        returns a detached tensor, with one RuntimeWarning per sampler instance; an input that requires grad is refused."""
        if torch.is_grad_enabled():
            if input_image.requires_grad or (y_T is not None and torch.is_tensor(y_T) and y_T.requires_grad):
                raise RuntimeError("GaussianDiffusionSampler.forward: an input requires grad, but the sampling loop runs under "
                                   "torch.no_grad() and cannot be differentiated; detach it or call under torch.no_grad()")
            if any(p.requires_grad for p in self.model.parameters()) and not getattr(self, "_warned_grad", False):
                self._warned_grad = True
                warnings.warn("GaussianDiffusionSampler.forward was called with autograd enabled: the sampling loop runs under "
                              "torch.no_grad() and returns a tensor without grad_fn", RuntimeWarning, stacklevel=2)
        with torch.no_grad():
            return self._forward(input_image, ddim, unconditional_guidance_scale, ddim_step, y_T, noise_by_step, trajectory)

    def _forward(self, input_image, ddim, unconditional_guidance_scale, ddim_step, y_T, noise_by_step, trajectory):
        if input_image.is_cuda and not input_image.is_contiguous():
            input_image = input_image.contiguous()
        E.require_gpu_tensor(input_image, "input_image")
        lib = _capi.lib()
        dev = input_image.device
        img = input_image.float() / 255.0                                                          # :220
        B, Cx, H, W = (int(v) for v in img.shape)
        if Cx != 3:
            raise RuntimeError(f"expected input[{B}, {Cx + 3}, {H}, {W}] to have 6 channels")
        if ddim and ddim_step is None:
            raise TypeError("unsupported operand type(s) for /: 'int' and 'NoneType'")             # :243 with ddim_step=None
        inject = (not ddim) and noise_by_step is not None
        seed = 0 if (ddim or inject) else int(torch.empty((), dtype=torch.int64).random_().item())
        key = (B, H, W, str(dev), int(ddim_step) if ddim else None, inject, seed, lib.hdiff_get_contraction_mode())
        sp = self._plans.get(key)
        if sp is None or sp.unet is not self.model.plan_for(B, H, W, dev, True):
            sp = _StepPlan(self, B, H, W, dev, int(ddim_step) if ddim else None, inject, seed)
            self._plans = {key: sp}                                  # a graph bakes its seed and contraction mode: keep one live plan
        sp.unet.plan.pack_weights()     # once per call: also catches writes through p.data, which p._version does not see
        self.model.dynamic_forward(torch.cat([img, img], dim=1))    # the reference runs it on every call (requires_grad only)
        y = torch.randn_like(img) if y_T is None else y_T            # :226 / :239
        up = sp.unet
        up.cond.copy_(img)
        up.y.copy_(y)
        sp.step.fill_(sp.n_steps - 1)
        sp.nan_flag.zero_()
        stream = torch.cuda.current_stream(dev).cuda_stream
        eager = inject or trajectory is not None or sp.n_steps < self.GRAPH_MIN_STEPS
        if not eager:
            sp.plan.capture()
        for k in range(sp.n_steps):
            if inject and k < sp.n_steps - 1:
                sp.noise.copy_(noise_by_step[k])
            if eager:
                sp.plan.run(stream)
            else:
                sp.plan.replay(stream)
            if trajectory is not None:
                trajectory.append(up.y.clone())
        out = torch.empty_like(img)
        _capi.check(lib.hdiff_clip(up.y.data_ptr(), out.data_ptr(), C.c_float(-1.0), C.c_float(1.0), sp.n, stream), "clip")
        return out
