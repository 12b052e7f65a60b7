"""hdiff_amd.optim.AdamW (csrc/optimizer.hip: the clip + AdamW tail of a training step as three launches over all tensors) against the calls
it replaces -- torch.nn.utils.clip_grad_norm_(params, max_norm); torch.optim.AdamW.step() (reference: TrainCondition.py:39-40, 61-63)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hdiff_amd  # noqa: E402
from hdiff_amd import optim as HO  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(1,), (5,), (3, 7), (4097,), (64, 3, 3, 3), (300001,), (2, 4096), (1 << 20,)]


def _pair(seed, views=False):
    g = torch.Generator().manual_seed(seed)
    a = [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).to(DEV)) for s in SHAPES]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    return a, b


@pytest.mark.parametrize("max_norm", [0.5, 1e9, None])
def test_adamw_with_fused_clip_matches_torch_over_several_steps(max_norm):
    """Five steps with fresh random gradients, a learning rate that changes between steps (the harness's schedulers write group["lr"]),
    tensors of 1 .. 2^20 elements (chunk tails, tensors smaller than a chunk).  max_norm 0.5 clips every step, 1e9 never does, None skips the
    norm.  Parameters and moments agree with torch's AdamW to fp32 rounding (the kernel follows torch's order of operations), the returned
    norm to 1e-6 relative, and the gradients are left clipped in place like clip_grad_norm_ leaves them."""
    pa, pb = _pair(0)
    ours = HO.AdamW(pa, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    ref = torch.optim.AdamW(pb, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1)
    for step in range(5):
        lr = 1e-3 * (1.0 + 0.5 * step)
        ours.param_groups[0]["lr"] = lr; ref.param_groups[0]["lr"] = lr
        for x, y in zip(pa, pb):
            gr = (torch.randn(x.shape, generator=g) * (10.0 ** (step - 2))).to(DEV)
            x.grad = gr.clone(); y.grad = gr.clone()
        if max_norm is not None:
            want = torch.nn.utils.clip_grad_norm_(pb, max_norm)
        ref.step()
        got = ours.step(max_grad_norm=max_norm)
        if max_norm is not None:
            assert abs(got.item() - want.item()) <= 1e-6 * want.item(), (step, got.item(), want.item())
            for x, y in zip(pa, pb):
                assert (x.grad - y.grad).abs().max().item() <= 2e-7 * max(1e-30, y.grad.abs().max().item()), step
        for x, y in zip(pa, pb):
            scale = max(1e-3, y.detach().abs().max().item())
            assert (x.detach() - y.detach()).abs().max().item() <= 1e-6 * scale, (step, x.shape)
            sx, sy = ours.state[x], ref.state[y]
            assert (sx["exp_avg"] - sy["exp_avg"]).abs().max().item() <= 1e-6 * max(1e-30, sy["exp_avg"].abs().max().item())
            assert (sx["exp_avg_sq"] - sy["exp_avg_sq"]).abs().max().item() <= 1e-6 * max(1e-30, sy["exp_avg_sq"].abs().max().item())
            assert int(sx["step"].item()) == step + 1


def test_adamw_on_flat_gradient_views_groups_and_missing_grads():
    """Gradients as views into one flat exchange buffer (parallel.FlatGradients: arbitrary 4-byte offsets), two parameter groups with their
    own lr / weight decay, one parameter that receives no gradient (skipped like torch skips it), bitwise repeatability of the whole step."""
    from hdiff_amd.parallel import FlatGradients

    def run():
        pa, pb = _pair(3)
        fl = FlatGradients(pa, world=1)
        fl.zero_()
        g = torch.Generator().manual_seed(4)
        for i, (x, y) in enumerate(zip(pa, pb)):
            if i == 2:
                x.grad = None; y.grad = None
                continue
            gr = torch.randn(x.shape, generator=g).to(DEV)
            x.grad.copy_(gr); y.grad = gr.clone()
        ours = HO.AdamW([{"params": pa[:4], "lr": 1e-3, "weight_decay": 0.0}, {"params": pa[4:], "lr": 5e-4}], weight_decay=1e-2)
        ref = torch.optim.AdamW([{"params": pb[:4], "lr": 1e-3, "weight_decay": 0.0}, {"params": pb[4:], "lr": 5e-4}], weight_decay=1e-2)
        want = torch.nn.utils.clip_grad_norm_([p for p in pb if p.grad is not None], 1.0)
        ref.step()
        got = ours.step(max_grad_norm=1.0)
        return pa, pb, got, want, fl
    pa, pb, got, want, _ = run()
    assert abs(got.item() - want.item()) <= 1e-6 * want.item()
    for i, (x, y) in enumerate(zip(pa, pb)):
        assert (x.detach() - y.detach()).abs().max().item() <= 1e-6 * max(1e-3, y.detach().abs().max().item()), i
    pa2, _, got2, _, _ = run()
    assert got2.item() == got.item() and all(torch.equal(x.detach(), z.detach()) for x, z in zip(pa, pa2)), "not bitwise reproducible"


def test_adamw_refuses_cpu_parameters_and_works_with_the_harness_schedulers():
    p = torch.nn.Parameter(torch.randn(8))
    p.grad = torch.randn(8)
    with pytest.raises(RuntimeError):
        HO.AdamW([p], lr=1e-3).step()
    from hdiff_amd.Scheduler import GradualWarmupScheduler
    q = torch.nn.Parameter(torch.randn(8, device=DEV))
    opt = HO.AdamW([q], lr=1e-4, weight_decay=1e-4)
    sched = GradualWarmupScheduler(optimizer=opt, multiplier=2.5, warm_epoch=2,
                                   after_scheduler=torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=opt, T_max=20, eta_min=0, last_epoch=-1))
    lrs = []
    for _ in range(4):
        q.grad = torch.ones_like(q)
        opt.step(max_grad_norm=1.0)
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
    assert lrs[0] > 1e-4 and len(set(lrs)) > 1
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}       # torch.optim.AdamW's state layout
