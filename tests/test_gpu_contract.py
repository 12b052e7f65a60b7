"""GPU tests of the drop-in contract's edges: what the reference does with bad indices, strided inputs, a changed guidance
weight, train-mode dropout without autograd, weights rewritten behind autograd's back -- and that a bad index can never fault
the GPU (the gather kernels clamp; the host raises what the reference raises).

Reference behaviour cited per test (paths relative to /root/reference/DiffusionFreeGuidence/).
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

import hdiff_amd  # noqa: E402
from hdiff_amd import _capi  # noqa: E402
from hdiff_amd.DiffusionFreeGuidence import DiffusionCondition as DC  # noqa: E402
from hdiff_amd.DiffusionFreeGuidence import ModelCondition as MC  # noqa: E402

DEV = "cuda:0"
SMALL = dict(T=8, num_labels=3, ch=32, ch_mult=[1, 2], num_res_blocks=1, dropout=0.0)


def small(seed=1, **over):
    torch.manual_seed(seed)
    return MC.UNet(**dict(SMALL, **over))


def test_gather_kernels_clamp_out_of_range_rows():
    """hdiff_linear_rows / _bwd / hdiff_q_sample called straight through the C ABI with indices outside the table: the
    result is that of the clamped index and the call returns cleanly (pins the fix of round 1's abort, where a recycled
    index buffer sent the gather to garbage rows).  Run once; nothing here tries to provoke a fault."""
    g = torch.Generator().manual_seed(1)
    n_rows, K, N = 20, 128, 64
    table = torch.randn(n_rows, K, generator=g).to(DEV)
    W, b = (torch.randn(N, K, generator=g) / 11).to(DEV), torch.randn(N, generator=g).to(DEV)
    bad = torch.tensor([-5, 3, 25, 2 ** 40, -2 ** 40, 19], dtype=torch.int64, device=DEV)
    good = bad.clamp(0, n_rows - 1)
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    y_bad, y_good = torch.empty(6, N, device=DEV), torch.empty(6, N, device=DEV)
    for idx, y in ((bad, y_bad), (good, y_good)):
        _capi.check(lib.hdiff_linear_rows(table.data_ptr(), idx.data_ptr(), n_rows, W.data_ptr(), b.data_ptr(), y.data_ptr(),
                                          6, K, N, 0, 0, s), "linear_rows")
    torch.cuda.synchronize()
    assert torch.equal(y_bad, y_good)
    assert (y_good - (table[good] @ W.t() + b)).abs().max().item() < 1e-4
    dy = torch.randn(6, N, generator=g).to(DEV)
    outs = []
    for idx in (bad, good):
        dx, dW, db = torch.zeros(n_rows, K, device=DEV), torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        _capi.check(lib.hdiff_linear_rows_bwd(table.data_ptr(), idx.data_ptr(), n_rows, W.data_ptr(), dy.data_ptr(),
                                              dx.data_ptr(), dW.data_ptr(), db.data_ptr(), 6, K, N, 0, 0, 0, s), "bwd")
        outs.append((dx, dW, db))
    torch.cuda.synchronize()
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    assert outs[0][0][0].abs().max().item() == 0.0            # pad_row = 0 gets no gradient although three indices clamp to it
    # q_sample: t outside [0, T)
    T, per = 10, 3 * 8 * 8
    sa, sb = torch.linspace(1, 0.1, T).to(DEV), torch.linspace(0, 0.9, T).to(DEV)
    x0, nz = torch.randn(3, per, generator=g).to(DEV), torch.randn(3, per, generator=g).to(DEV)
    t_bad = torch.tensor([-1, T + 5, 4], dtype=torch.int64, device=DEV)
    out = torch.empty(3, per, device=DEV)
    _capi.check(lib.hdiff_q_sample(x0.data_ptr(), nz.data_ptr(), t_bad.data_ptr(), sa.data_ptr(), sb.data_ptr(),
                                   out.data_ptr(), 3, per, T, s), "q_sample")
    tc = t_bad.clamp(0, T - 1)
    assert torch.equal(out, sa[tc][:, None] * x0 + sb[tc][:, None] * nz)


def test_bad_indices_raise_like_the_reference():
    """nn.Embedding raises IndexError for t >= T or labels > num_labels (ModelCondition.py:38,56 -> 257-258); extract's
    torch.gather raises RuntimeError for a time step outside the schedule (DiffusionCondition.py:13)."""
    m = small().to(DEV).eval()
    x = torch.randn(2, 3, 16, 16, device=DEV)
    ok_t, ok_l = torch.tensor([0, 7], device=DEV), torch.tensor([3, 0], device=DEV)
    with torch.no_grad():
        m(x, ok_t, ok_l)
        for t, lab in ((torch.tensor([0, 8], device=DEV), ok_l), (torch.tensor([-1, 2], device=DEV), ok_l),
                       (ok_t, torch.tensor([4, 0], device=DEV)), (ok_t, torch.tensor([1, -1], device=DEV))):
            with pytest.raises(IndexError):
                m(x, t, lab)
        assert torch.equal(m(x, ok_t.to(torch.int32), ok_l.to(torch.int32)), m(x, ok_t, ok_l))     # int32 indices as nn.Embedding
    mt = small().to(DEV).train()
    with pytest.raises(IndexError):                            # the training (autograd) path validates too
        mt(x, torch.tensor([0, 8], device=DEV), ok_l)
    tr = DC.GaussianDiffusionTrainer(mt, 1e-4, 0.02, 8).to(DEV)
    with pytest.raises(RuntimeError, match="out of bounds"):
        tr(x, ok_l, t=torch.tensor([0, 8], device=DEV))      # raised before any kernel sees the index
    samp = DC.GaussianDiffusionSampler(m, 1e-4, 0.02, 8, w=1.0).to(DEV)
    with torch.no_grad(), pytest.raises(RuntimeError, match="out of bounds"):
        samp.predict_xt_prev_mean_from_eps(x, torch.tensor([8, 0], device=DEV), x)
    with torch.no_grad(), pytest.raises(RuntimeError):
        samp.predict_xt_prev_mean_from_eps(x, torch.tensor([1.0, 0.0], device=DEV), x)               # float index: gather refuses
    with torch.no_grad(), pytest.raises(IndexError):
        samp(x, torch.tensor([9, 0], device=DEV))


def test_sampler_reads_w_and_weights_at_call_time():
    """The reference evaluates self.w on every step (DiffusionCondition.py:78) and reads the live weights; a captured step
    must follow a changed sampler.w, and a write through p.data (which autograd's version counter does not see)."""
    m = small().to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    x_T = torch.randn(2, 3, 16, 16, generator=g).to(DEV)
    lab = torch.tensor([1, 2], device=DEV)
    z = torch.randn(8, 2, 3, 16, 16, generator=g).to(DEV)
    with torch.no_grad():
        s0 = DC.GaussianDiffusionSampler(m, 1e-4, 0.028, 8, w=0.0).to(DEV)
        a0 = s0(x_T, lab, noise_by_step=z)
        s0.w = 1.8
        a18 = s0(x_T, lab, noise_by_step=z)
        fresh = DC.GaussianDiffusionSampler(m, 1e-4, 0.028, 8, w=1.8).to(DEV)(x_T, lab, noise_by_step=z)
        assert not torch.equal(a0, a18) and torch.equal(a18, fresh)
        # weights rewritten through .data after a plan exists
        m.tail[2].weight.data.mul_(0.5)
        m.downblocks[0].attn.in_proj_weight.data.add_(0.01)
        b = s0(x_T, lab, noise_by_step=z)
        m2 = small().to(DEV).eval()
        m2.load_state_dict(m.state_dict())
        want = DC.GaussianDiffusionSampler(m2, 1e-4, 0.028, 8, w=1.8).to(DEV)(x_T, lab, noise_by_step=z)
        assert not torch.equal(b, a18) and torch.equal(b, want)
        # UNet.forward alone: explicit invalidation for .data writes, automatic for versioned writes
        t = torch.tensor([3, 4], device=DEV)
        y0 = m(x_T, t, lab)
        m.head.bias.data.add_(0.25)
        m.invalidate_packed()
        y1 = m(x_T, t, lab)
        assert not torch.equal(y0, y1)
        m.head.weight.mul_(1.5)                                  # versioned in-place write: seen without any call
        assert torch.equal(m(x_T, t, lab), m2_like(m, x_T, t, lab))


def m2_like(m, x, t, lab):
    m2 = small().to(DEV).eval()
    m2.load_state_dict(m.state_dict())
    with torch.no_grad():
        return m2(x, t, lab)


def test_strided_inputs_are_accepted():
    m = small().to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    big = torch.randn(2, 3, 32, 32, generator=g).to(DEV)
    x = big[:, :, ::2, ::2]                                      # a strided view, as the reference's ATen ops accept
    assert not x.is_contiguous()
    t, lab = torch.tensor([1, 5], device=DEV), torch.tensor([2, 0], device=DEV)
    tt = torch.stack([t, t], dim=1)[:, 0]                        # strided index vector
    with torch.no_grad():
        assert torch.equal(m(x, tt, lab), m(x.contiguous(), t, lab))
        samp = DC.GaussianDiffusionSampler(m, 1e-4, 0.028, 8, w=1.0).to(DEV)
        z = torch.randn(8, 2, 3, 16, 16, generator=g).to(DEV)
        assert torch.equal(samp(x, lab, noise_by_step=z), samp(x.contiguous(), lab, noise_by_step=z))
        tr = DC.GaussianDiffusionTrainer(m, 1e-4, 0.028, 8).to(DEV)
        nz = torch.randn(2, 3, 16, 16, generator=g).to(DEV)
        assert torch.equal(tr(x, lab, t=t, noise=nz), tr(x.contiguous(), lab, t=t, noise=nz))


def test_train_mode_dropout_without_autograd():
    """nn.Dropout acts in train mode whether or not autograd records (ModelCondition.py:185): under torch.no_grad() a
    train-mode model must run (with dropout), not raise."""
    m = small(dropout=0.3).to(DEV)
    x = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(5)).to(DEV)
    t, lab = torch.tensor([1, 5], device=DEV), torch.tensor([2, 0], device=DEV)
    with torch.no_grad():
        m.eval()
        y_eval = m(x, t, lab)
        m.train()
        torch.manual_seed(10); y1 = m(x, t, lab)
        torch.manual_seed(10); y2 = m(x, t, lab)
        torch.manual_seed(11); y3 = m(x, t, lab)
        assert torch.isfinite(y1).all() and torch.equal(y1, y2)
        assert not torch.equal(y1, y3) and not torch.equal(y1, y_eval)
        # a ResBlock on its own, train mode
        rb = MC.ResBlock(32, 32, 64, 0.5, attn=False).to(DEV).train()
        h, temb, cemb = torch.randn(2, 32, 8, 8, device=DEV), torch.randn(2, 64, device=DEV), torch.randn(2, 64, device=DEV)
        torch.manual_seed(1); r1 = rb(h, temb, cemb)
        torch.manual_seed(1); r2 = rb(h, temb, cemb)
        assert torch.equal(r1, r2) and not torch.equal(r1, rb.eval()(h, temb, cemb))


def test_plane_count_beyond_65535():
    """B * C >= 65 536 planes (a batch of 128 with the 512-channel concat blocks): the apply kernels index planes on
    gridDim.x, not on the 65 535-limited y."""
    B, Cc, HW, G = 130, 512, 16, 32
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(B, Cc, HW, device=DEV, generator=g)
    scale, shift = torch.rand(B, Cc, device=DEV, generator=g) + 0.5, torch.randn(B, Cc, device=DEV, generator=g)
    y = torch.empty_like(x)
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    _capi.check(lib.hdiff_gn_swish_apply(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), B, Cc, HW, s), "apply")
    a = x * scale[:, :, None] + shift[:, :, None]
    assert (y - a * torch.sigmoid(a)).abs().max().item() < 1e-5
    # backward through the autograd wrapper's kernel (hdiff_gn_swish_bwd) against torch autograd in float64
    gamma, beta = torch.rand(Cc, device=DEV, generator=g) + 0.5, torch.randn(Cc, device=DEV, generator=g)
    dA = torch.randn(B, Cc, HW, device=DEV, generator=g)
    x64 = x.double().requires_grad_(True)
    a64 = torch.nn.functional.group_norm(x64, G, gamma.double(), beta.double(), 1e-5)
    (a64 * torch.sigmoid(a64)).backward(dA.double())
    xr = x.view(B, G, -1)
    mean = xr.mean(-1)
    rstd = 1.0 / torch.sqrt(xr.var(-1, unbiased=False) + 1e-5)
    ws = torch.empty(2 * B * Cc + 2 * B * G, device=DEV)
    dx, dgw, dgb = torch.empty_like(x), torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    _capi.check(lib.hdiff_gn_swish_bwd(x.data_ptr(), None, Cc, 0, B, HW, G, dA.data_ptr(), mean.contiguous().data_ptr(),
                                       rstd.contiguous().data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(),
                                       dx.data_ptr(), None, dgw.data_ptr(), dgb.data_ptr(), s), "gn_swish_bwd")
    assert (dx.double() - x64.grad).abs().max().item() < 1e-4 * x64.grad.abs().max().item()


def test_split_bf16_conv_refuses_a_permuted_or_repeated_tap_list():
    """include/hdiff.h (wp_x3): hdiff_pack_conv_weight_x3 stores tap t = (t / 3, t % 3).  A descriptor that lists the nine
    taps in another order (its fp32 pack wp follows that order) must not be fed to the split-bf16 kernel, which would pair
    the descriptor's offsets with the canonical pack -- it keeps the fp32 kernel, which reads the order from the descriptor.
    Same for a list with a repeated tap.  (conv_igemm.hip is_x3_conv.)"""
    import math
    import torch.nn.functional as F
    from hdiff_amd import engine as E
    hdiff_amd.set_contraction_mode("bf16x3")
    try:
        g = torch.Generator().manual_seed(3)
        B, Cin, Cout, H, W = 16, 64, 64, 64, 64                       # large enough for the split-bf16 kernel (>= 192 workgroups)
        x = torch.randn(B, Cin, H, W, generator=g).cuda()
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).cuda()
        want = F.conv2d(x.double(), w.double(), padding=1)

        def run(perm, repeat=False):
            plan = E.Plan("cuda")
            taps = E.conv_taps(3, 1)
            taps = E.TapSet([taps.dy[i] for i in perm], [taps.dx[i] for i in perm], [taps.ky[i] for i in perm], [taps.kx[i] for i in perm])
            if repeat:                                               # tap 0 twice, the last one dropped: some other convolution
                taps.dy[8], taps.dx[8], taps.ky[8], taps.kx[8] = taps.dy[0], taps.dx[0], taps.ky[0], taps.kx[0]
            pk = E._new_pack(plan, Cout, Cin, taps)
            pk.add_source(w, 0, taps.ky, taps.kx, 0)
            pk.enable_x3(w)                                          # canonical three-piece pack, whatever the list says
            out = plan.buf(B, Cout, H, W)
            plan.conv(x, None, pk, None, out, B=B, H=H, W=W, VH=H, VW=W)
            plan.pack_weights()
            plan.run()
            torch.cuda.synchronize()
            return out.clone()

        canon = run(list(range(9)))
        perm = run([8, 0, 3, 1, 7, 2, 6, 4, 5])
        err = lambda o: (o.double() - want).abs().max().item()
        assert err(canon) < 2e-5 and err(perm) < 2e-5, (err(canon), err(perm))
        assert not torch.equal(canon, perm), "the permuted list should have run the fp32-input kernel, not the split-bf16 one"
        rep = run(list(range(9)), repeat=True)
        w_rep = w.clone()
        w_rep[:, :, 2, 2] = 0
        w_rep_extra = torch.zeros_like(w)
        w_rep_extra[:, :, 0, 0] = w[:, :, 0, 0]
        want_rep = F.conv2d(x.double(), (w_rep + w_rep_extra).double(), padding=1)
        assert (rep.double() - want_rep).abs().max().item() < 2e-5
    finally:
        hdiff_amd.set_contraction_mode("bf16x3")


def test_sampler_runs_with_autograd_enabled_like_the_reference():
    """The reference's sampler can be called with autograd on (DiffusionCondition.py:82-98; it then records a graph nobody
    differentiates).  Here the loop enters torch.no_grad() itself: same values as under no_grad, a detached result, one
    RuntimeWarning per sampler INSTANCE (a second sampler warns again: ADVICE round 4), and an x_T that requires grad -- a caller
    who expects that graph -- is refused."""
    import warnings
    m = small(seed=4).cuda().eval()
    cfg = SMALL
    s = DC.GaussianDiffusionSampler(m, 1e-4, 0.02, cfg["T"], w=1.2).cuda()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 16, 16, generator=g).cuda()
    lab = torch.tensor([1, 2]).cuda()
    noise = torch.randn(cfg["T"], 2, 3, 16, 16, generator=g).cuda()
    with torch.no_grad():
        want = s(x, lab, noise_by_step=noise)
    s2 = DC.GaussianDiffusionSampler(m, 1e-4, 0.02, cfg["T"], w=1.2).cuda()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        assert torch.is_grad_enabled() and any(p.requires_grad for p in m.parameters())
        got = s(x, lab, noise_by_step=noise)
        again = s(x, lab, noise_by_step=noise)
        other = s2(x, lab, noise_by_step=noise)
    assert torch.equal(got, want) and torch.equal(again, want) and torch.equal(other, want)
    assert not got.requires_grad and got.grad_fn is None
    with pytest.raises(RuntimeError, match="requires grad"):
        s(x.clone().requires_grad_(True), lab, noise_by_step=noise)
    assert sum(issubclass(r.category, RuntimeWarning) and "autograd enabled" in str(r.message) for r in rec) == 2      # once per instance


def test_reserve_split_workspace_opt_out():
    """plans reserve the split-operand attention scratch whatever the mode (18 bytes per qkv element).  An f32-only deployment
    opts out with hdiff_amd.reserve_split_workspace(False) BEFORE building its plans: the attention calls then carry no workspace, the plan
    holds less memory, and a run in the split-operand mode is still correct (the library takes the in-loop-split kernel for a null pointer)."""
    import hdiff_amd
    from hdiff_amd import engine
    from hdiff_amd.DiffusionFreeGuidence.ModelCondition import UNet
    cfg = dict(T=4, num_labels=2, ch=128, ch_mult=[1, 1], num_res_blocks=1, dropout=0.0)      # heads of 16 channels at L = 1024: the pre-split kernels' shapes
    torch.manual_seed(0)
    m = UNet(**cfg).eval().to(DEV)
    x = torch.randn(2, 3, 32, 32, device=DEV); t = torch.tensor([1, 3], device=DEV); y = torch.tensor([1, 2], device=DEV)
    before = hdiff_amd.get_contraction_mode()
    try:
        hdiff_amd.set_contraction_mode("bf16x3")
        with torch.no_grad():
            ref = m(x, t, y).clone()
        with_ws = m.plan_for(2, 32, 32, torch.device(DEV)).plan
        att = [a for n, _, a in with_ws.ops if n == "hdiff_mha_flash_fwd_ws"]
        assert att and any(a[7] is not None for a in att)
        bytes_with = with_ws.bytes_allocated()
        hdiff_amd.reserve_split_workspace(False)
        m2 = UNet(**cfg).eval().to(DEV)
        m2.load_state_dict(m.state_dict())
        with torch.no_grad():
            out = m2(x, t, y)
        lean = m2.plan_for(2, 32, 32, torch.device(DEV)).plan
        assert all(a[7] is None and a[8].value == 0 for n, _, a in lean.ops if n == "hdiff_mha_flash_fwd_ws")
        assert lean.bytes_allocated() < bytes_with
        assert (out - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())      # another kernel of the same error class
    finally:
        hdiff_amd.reserve_split_workspace(True)
        hdiff_amd.set_contraction_mode(before)
        assert engine.RESERVE_SPLIT_WORKSPACE is True
