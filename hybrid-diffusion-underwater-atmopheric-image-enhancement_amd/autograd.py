"""Training (autograd) path of the HIP kernels: the reference's ``loss.backward()`` (TrainCondition.py:59-60) through
``GaussianDiffusionTrainer.forward`` -> ``UNet.forward`` runs on hand-written gfx950 kernels in both directions.

Each ``torch.autograd.Function`` below is one fused launch group of the forward (the same groups the inference planner
emits) with its hand-written backward:

  _FusedConv   GroupNorm-Swish prologue + conv (3x3 / 1x1) + bias/+vec/+residual epilogue
               bwd: dgrad = the forward kernel on flipped, transposed packed weights; wgrad = conv_wgrad.hip (prologue
               recomputed on the fly); GroupNorm-Swish bwd = groupnorm_bwd.hip; bias / vec grads = plane reductions
  _DownFn      DownSample (3x3/s2 + 5x5/s2 folded into one 5x5/s2); dgrad = 4 transposed-conv phases
  _TConvFn     ConvTranspose2d(5, s2) as 4 phases; dgrad = one 5x5/s2 conv
  _FlashFn     flash attention core; bwd recomputes P from the saved log-sum-exp (attention_bwd.hip)
  _LinearFn / _VecFn   the small dense layers (time / label MLPs, per-block projections)
  _SqErrFn     the unreduced squared error

PyTorch provides the autograd graph, gradient accumulation and the optimizer; no arithmetic of the model runs in ATen.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from . import _capi
from . import engine as E

GN_GROUPS, GN_EPS, NUM_HEADS = E.GN_GROUPS, E.GN_EPS, E.NUM_HEADS


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.contiguous()


# ----------------------------------------------------------------------------------------------------------------------
# launch helpers (immediate execution)
# ----------------------------------------------------------------------------------------------------------------------
def _run_conv(x0, x1, sources, taps: E.TapSet, cout: int, cin: int, bias, out, *, B, H, W, VH, VW, in_stride=1,
              out_map=(1, 0, 1, 0), gn=None, addvec=None, residual=None, x3=None, act_range=None) -> None:
    """x3 = (forward weight [3x3], transposed?): also pack the split-bf16 copy, so that a plain 3x3 / stride-1 launch (the
    forward conv, or its input-gradient conv on the transposed, mirrored weight) runs conv3x3_x3.hip in the bf16x3 mode.
    act_range = (gamma, beta, group_elems, gain) of the GroupNorm + Swish the input went through (Plan.conv): the forward conv
    then takes that kernel's fp16-pair form."""
    plan = E.Plan(x0.device)
    pk = E.PackedConv(x0.device, cout, cin, taps)
    for (w, mode, ky, kx, acc) in sources:
        pk.add_source(w, mode, ky, kx, acc)
    if x3 is not None and in_stride == 1 and out_map == (1, 0, 1, 0):
        if len(taps.dy) == 1:
            pk.enable_x3_taps(x3[0], 1 if x3[1] else 0)      # 1x1: one tap on bf16 triples (conv1x1_x3.hip); transposed = the [in][out] reading
        else:
            pk.enable_x3(x3[0], transposed=x3[1])
    plan.packs.append(pk)
    plan.conv(x0, x1, pk, bias, out, B=B, H=H, W=W, VH=VH, VW=VW, in_stride=in_stride, out_map=out_map, gn=gn,
              addvec=addvec, residual=residual, act_range=act_range)
    plan.pack_weights()
    plan.run()


def _run_wgrad(x0, x1, gn, dy, taps: E.TapSet, cout: int, cin: int, *, B, H, W, VH, VW, in_stride=1, out_map=(1, 0, 1, 0),
               targets: Sequence[Tuple[torch.Tensor, int, Sequence[int], Sequence[int], int]]) -> None:
    """targets: (dw tensor in PyTorch layout, mode, tap_ky, tap_kx, accumulate)."""
    lib = _capi.lib()
    dev = x0.device
    d = _capi.WgradDesc()
    C0 = int(x0.shape[1])
    C1 = int(x1.shape[1]) if x1 is not None else 0
    assert C0 + C1 == cin
    d.x0, d.x1, d.C0, d.C1, d.B, d.H, d.W = _p(x0), _p(x1), C0, C1, B, H, W
    d.gn_scale, d.gn_shift = (_p(gn[0]), _p(gn[1])) if gn is not None else (None, None)
    d.dy, d.Cout, d.CinPad, d.CoutPad = dy.data_ptr(), cout, E._pad(cin, 8), E._pad(cout, 64)
    d.OH, d.OW, d.VH, d.VW, d.in_stride = int(dy.shape[2]), int(dy.shape[3]), VH, VW, in_stride
    d.out_sy, d.out_oy, d.out_sx, d.out_ox = out_map
    n = len(taps.dy)
    d.ntaps = n
    for i in range(n):
        d.tap_dy[i], d.tap_dx[i] = taps.dy[i], taps.dx[i]
    nsplit, nfloats = C.c_int(0), C.c_int64(0)
    _capi.check(lib.hdiff_conv2d_wgrad_workspace(C.byref(d), C.byref(nsplit), C.byref(nfloats)), "wgrad_workspace")
    ws = torch.empty(nfloats.value, dtype=torch.float32, device=dev)
    s = _stream(dev)
    _capi.check(lib.hdiff_conv2d_wgrad(C.byref(d), ws.data_ptr(), nsplit.value, s), "conv2d_wgrad")
    for dw, mode, ky, kx, acc in targets:
        kh, kw = int(dw.shape[2]), int(dw.shape[3])
        a_ky, a_kx = (C.c_int * n)(*ky), (C.c_int * n)(*kx)
        _capi.check(lib.hdiff_conv_wgrad_unpack(ws.data_ptr(), nsplit.value, dw.data_ptr(), mode, cout, cin, kh, kw, n, a_ky,
                                                a_kx, d.CinPad, d.CoutPad, acc, s), "wgrad_unpack")


def _gn_stats(x0, x1, gamma, beta, B, HW):
    """-> (scale [B,C], shift [B,C], mean [B,G], rstd [B,G])"""
    lib = _capi.lib()
    dev = x0.device
    C0 = int(x0.shape[1])
    C1 = int(x1.shape[1]) if x1 is not None else 0
    Ct = C0 + C1
    if Ct % GN_GROUPS != 0:
        raise RuntimeError(f"Expected number of channels in input to be divisible by num_groups, got {Ct}")
    nsplit = max(1, min(32, HW // 4096))     # a function of the plane size only: per-sample results must not depend on B
    ws = torch.empty(B * GN_GROUPS * nsplit * 3, device=dev)
    scale, shift = torch.empty(B, Ct, device=dev), torch.empty(B, Ct, device=dev)
    mean, rstd = torch.empty(B, GN_GROUPS, device=dev), torch.empty(B, GN_GROUPS, device=dev)
    s = _stream(dev)
    _capi.check(lib.hdiff_gn_stats(_p(x0), _p(x1), C0, C1, B, HW, GN_GROUPS, nsplit, ws.data_ptr(), s), "gn_stats")
    _capi.check(lib.hdiff_gn_finalize(ws.data_ptr(), B, Ct, GN_GROUPS, nsplit, gamma.data_ptr(), beta.data_ptr(),
                                      C.c_float(GN_EPS), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                      rstd.data_ptr(), s), "gn_finalize")
    return scale, shift, mean, rstd


def _plane_sums(dy, want_vec: bool, want_bias: bool):
    lib = _capi.lib()
    B, Cc = int(dy.shape[0]), int(dy.shape[1])
    HW = dy.numel() // (B * Cc)
    dvec = torch.empty(B, Cc, device=dy.device) if (want_vec or want_bias) else None   # also the scratch of the bias sum
    dbias = torch.empty(Cc, device=dy.device) if want_bias else None
    if want_vec or want_bias:
        _capi.check(lib.hdiff_bias_addvec_grad(dy.data_ptr(), B, Cc, HW, _p(dvec), _p(dbias), _stream(dy.device)),
                    "bias_addvec_grad")
    return (dvec if want_vec else None), dbias


# ----------------------------------------------------------------------------------------------------------------------
# fused conv
# ----------------------------------------------------------------------------------------------------------------------
class _FusedConv(Function):
    @staticmethod
    def forward(ctx, x0, x1, weight, bias, gn_w, gn_b, addvec, residual, k: int, drop_p: float, seed: int):
        x0, x1, addvec, residual = _c(x0), _c(x1), _c(addvec), _c(residual)
        weight = weight.contiguous()
        lib = _capi.lib()
        dev = x0.device
        B, C0, H, W = (int(v) for v in x0.shape)
        C1 = int(x1.shape[1]) if x1 is not None else 0
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        assert cin == C0 + C1
        pad = k // 2
        gn = mean = rstd = mask = act_range = None
        conv_in0, conv_in1, conv_gn = x0, x1, None
        if gn_w is not None:
            scale, shift, mean, rstd = _gn_stats(x0, x1, gn_w, gn_b, B, H * W)
            gn = (scale, shift)
            conv_gn = gn
            # the conv's input is swish(GroupNorm(x)) (times mask / keep behind the dropout): its range follows from the
            # GroupNorm weights, which is what the fp16-pair 3x3 kernel stages its activations by
            act_range = (gn_w.detach(), gn_b.detach(), (cin // GN_GROUPS) * H * W, 1.0 / (1.0 - drop_p) if drop_p > 0.0 else 1.0)
            if drop_p > 0.0:
                # nn.Dropout sits between Swish and the conv (ModelCondition.py:185): materialise a = swish(gn(x)) * mask
                assert x1 is None
                a = torch.empty_like(x0)
                s = _stream(dev)
                _capi.check(lib.hdiff_gn_swish_apply(x0.data_ptr(), scale.data_ptr(), shift.data_ptr(), a.data_ptr(), B, C0,
                                                     H * W, s), "gn_swish_apply")
                mask = torch.empty_like(x0)
                _capi.check(lib.hdiff_dropout_mask(mask.data_ptr(), mask.numel(), C.c_float(1.0 - drop_p), C.c_uint64(seed),
                                                   C.c_uint64(0), s), "dropout_mask")
                _capi.check(lib.hdiff_mul(a.data_ptr(), mask.data_ptr(), a.data_ptr(), a.numel(), s), "mul")
                conv_in0, conv_in1, conv_gn = a, None, None
        taps = E.conv_taps(k, pad)
        out = torch.empty(B, cout, H, W, device=dev)
        _run_conv(conv_in0, conv_in1, [(weight, 0, taps.ky, taps.kx, 0)], taps, cout, cin, bias, out, B=B, H=H, W=W, VH=H,
                  VW=W, gn=conv_gn, addvec=addvec, residual=residual, x3=(weight, False) if k in (1, 3) else None,
                  act_range=act_range if k == 3 else None)
        ctx.k, ctx.has_x1, ctx.has_gn, ctx.dropped = k, x1 is not None, gn_w is not None, mask is not None
        ctx.has_bias, ctx.has_vec, ctx.has_res = bias is not None, addvec is not None, residual is not None
        saved = [x0, x1, weight, gn_w, gn_b, mean, rstd, gn[0] if gn else None, gn[1] if gn else None, mask,
                 conv_in0 if mask is not None else None]
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        x0, x1, weight, gn_w, gn_b, mean, rstd, scale, shift, mask, a_masked = ctx.saved_tensors
        dout = dout.contiguous()
        lib = _capi.lib()
        dev = dout.device
        B, C0, H, W = (int(v) for v in x0.shape)
        C1 = int(x1.shape[1]) if x1 is not None else 0
        cout, cin = int(weight.shape[0]), int(weight.shape[1])
        k, pad = ctx.k, ctx.k // 2
        need = ctx.needs_input_grad
        taps = E.conv_taps(k, pad)
        s = _stream(dev)

        d_res = dout if (ctx.has_res and need[7]) else None
        d_vec, d_bias = _plane_sums(dout, ctx.has_vec and need[6], ctx.has_bias and need[3])

        d_w = None
        if need[2]:
            d_w = torch.empty_like(weight)
            if ctx.dropped:
                _run_wgrad(a_masked, None, None, dout, taps, cout, cin, B=B, H=H, W=W, VH=H, VW=W,
                           targets=[(d_w, 0, taps.ky, taps.kx, 0)])
            else:
                _run_wgrad(x0, x1, (scale, shift) if ctx.has_gn else None, dout, taps, cout, cin, B=B, H=H, W=W, VH=H, VW=W,
                           targets=[(d_w, 0, taps.ky, taps.kx, 0)])

        d_x0 = d_x1 = d_gw = d_gb = None
        need_dx = need[0] or (ctx.has_x1 and need[1]) or (ctx.has_gn and (need[4] or need[5]))
        if need_dx:
            # dgrad: the forward kernel with flipped taps and the weight read as [GEMM-in = Cout][GEMM-out = Cin]
            # taps listed in the forward's (dy, dx) order, each reading the mirrored kernel element: the same set of products
            # as [pad - ky], and the order the specialised 3x3 kernels recognise
            dtaps = E.TapSet(taps.dy, taps.dx, [pad - dy for dy in taps.dy], [pad - dx for dx in taps.dx])
            dA = torch.empty(B, cin, H, W, device=dev)
            _run_conv(dout, None, [(weight, 1, dtaps.ky, dtaps.kx, 0)], dtaps, cin, cout, None, dA, B=B, H=H, W=W, VH=H, VW=W,
                      x3=(weight, True) if k in (1, 3) else None)
            if ctx.dropped:
                _capi.check(lib.hdiff_mul(dA.data_ptr(), mask.data_ptr(), dA.data_ptr(), dA.numel(), s), "mul")
            if ctx.has_gn:
                d_x0 = torch.empty_like(x0)
                d_x1 = torch.empty_like(x1) if x1 is not None else None
                d_gw, d_gb = torch.empty_like(gn_w), torch.empty_like(gn_b)
                ws = torch.empty(2 * B * cin + 2 * B * GN_GROUPS, device=dev)
                _capi.check(lib.hdiff_gn_swish_bwd(x0.data_ptr(), _p(x1), C0, C1, B, H * W, GN_GROUPS, dA.data_ptr(),
                                                   mean.data_ptr(), rstd.data_ptr(), gn_w.data_ptr(), gn_b.data_ptr(),
                                                   ws.data_ptr(), d_x0.data_ptr(), _p(d_x1), d_gw.data_ptr(),
                                                   d_gb.data_ptr(), s), "gn_swish_bwd")
            elif x1 is not None:
                d_x0, d_x1 = dA[:, :C0], dA[:, C0:]
            else:
                d_x0 = dA
        return d_x0, d_x1, d_w, d_bias, d_gw, d_gb, d_vec, d_res, None, None, None


# ----------------------------------------------------------------------------------------------------------------------
# DownSample / transposed conv
# ----------------------------------------------------------------------------------------------------------------------
def _down_center_map(taps: E.TapSet):
    ky3 = [ky - 1 if (1 <= ky <= 3 and 1 <= kx <= 3) else -1 for ky, kx in zip(taps.ky, taps.kx)]
    kx3 = [kx - 1 if (1 <= ky <= 3 and 1 <= kx <= 3) else 0 for ky, kx in zip(taps.ky, taps.kx)]
    return ky3, kx3


class _DownFn(Function):
    """c1(x) + c2(x) (ModelCondition.py:74-76) as one 5x5/s2 conv with the 3x3 folded into the 5x5 centre."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x, w1, w2 = x.contiguous(), w1.contiguous(), w2.contiguous()
        lib = _capi.lib()
        B, Cc, H, W = (int(v) for v in x.shape)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        taps = E.conv_taps(5, 2)
        ky3, kx3 = _down_center_map(taps)
        bias = torch.empty_like(b1)
        _capi.check(lib.hdiff_axpby(C.c_float(1.0), b1.data_ptr(), C.c_float(1.0), b2.data_ptr(), bias.data_ptr(), Cc,
                                    _stream(x.device)), "axpby")
        out = torch.empty(B, Cc, OH, OW, device=x.device)
        _run_conv(x, None, [(w2, 0, taps.ky, taps.kx, 0), (w1, 0, ky3, kx3, 1)], taps, Cc, Cc, bias, out, B=B, H=H, W=W,
                  VH=OH, VW=OW, in_stride=2)
        ctx.save_for_backward(x, w1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, w2 = ctx.saved_tensors
        dout = dout.contiguous()
        B, Cc, H, W = (int(v) for v in x.shape)
        OH, OW = int(dout.shape[2]), int(dout.shape[3])
        taps = E.conv_taps(5, 2)
        ky3, kx3 = _down_center_map(taps)
        _, d_b = _plane_sums(dout, False, True)
        d_w1, d_w2 = torch.empty_like(w1), torch.empty_like(w2)
        _run_wgrad(x, None, None, dout, taps, Cc, Cc, B=B, H=H, W=W, VH=OH, VW=OW, in_stride=2,
                   targets=[(d_w2, 0, taps.ky, taps.kx, 0), (d_w1, 0, ky3, kx3, 0)])
        d_x = None
        if ctx.needs_input_grad[0]:
            # dX = transposed conv of dY with the folded weights: 4 output-parity phases
            d_x = torch.zeros_like(x) if (H % 2 or W % 2) else torch.empty_like(x)
            for py in (0, 1):
                for px in (0, 1):
                    pt = E.tconv_phase_taps(py, px)
                    k3y = [ky - 1 if (1 <= ky <= 3 and 1 <= kx <= 3) else -1 for ky, kx in zip(pt.ky, pt.kx)]
                    k3x = [kx - 1 if (1 <= ky <= 3 and 1 <= kx <= 3) else 0 for ky, kx in zip(pt.ky, pt.kx)]
                    VH, VW = (H - py + 1) // 2, (W - px + 1) // 2
                    if VH <= 0 or VW <= 0:
                        continue
                    _run_conv(dout, None, [(w2, 1, pt.ky, pt.kx, 0), (w1, 1, k3y, k3x, 1)], pt, Cc, Cc, None, d_x, B=B, H=OH,
                              W=OW, VH=VH, VW=VW, out_map=(2, py, 2, px))
        return d_x, d_w1, d_b, d_w2, d_b.clone()


class _TConvFn(Function):
    """ConvTranspose2d(C, C, 5, 2, 2, 1) (ModelCondition.py:83) as 4 output-parity phases of stride-1 convs."""

    @staticmethod
    def forward(ctx, x, wt, bt):
        x, wt = x.contiguous(), wt.contiguous()
        B, Cc, H, W = (int(v) for v in x.shape)
        cout = int(wt.shape[1])
        u = torch.empty(B, cout, 2 * H, 2 * W, device=x.device)
        for py in (0, 1):
            for px in (0, 1):
                pt = E.tconv_phase_taps(py, px)
                _run_conv(x, None, [(wt, 1, pt.ky, pt.kx, 0)], pt, cout, Cc, bt, u, B=B, H=H, W=W, VH=H, VW=W,
                          out_map=(2, py, 2, px))
        ctx.save_for_backward(x, wt)
        return u

    @staticmethod
    def backward(ctx, du):
        x, wt = ctx.saved_tensors
        du = du.contiguous()
        B, Cc, H, W = (int(v) for v in x.shape)
        cout = int(wt.shape[1])
        _, d_b = _plane_sums(du, False, True)
        d_wt = torch.empty_like(wt)
        for py in (0, 1):
            for px in (0, 1):
                pt = E.tconv_phase_taps(py, px)
                _run_wgrad(x, None, None, du, pt, cout, Cc, B=B, H=H, W=W, VH=H, VW=W, out_map=(2, py, 2, px),
                           targets=[(d_wt, 1, pt.ky, pt.kx, 0)])
        d_x = None
        if ctx.needs_input_grad[0]:
            # dX[iy] = sum_k dU[2*iy - 2 + k] * Wt[k]: a 5x5 stride-2 conv over dU with the weight read as [out=Cin][in=Cout]
            taps = E.conv_taps(5, 2)
            d_x = torch.empty_like(x)
            # weight layout [Cin][Cout][5][5] == [GEMM-out][GEMM-in][ky][kx] == mode 0
            _run_conv(du, None, [(wt, 0, taps.ky, taps.kx, 0)], taps, Cc, cout, None, d_x, B=B, H=2 * H, W=2 * W, VH=H, VW=W,
                      in_stride=2)
        return d_x, d_wt, d_b


# ----------------------------------------------------------------------------------------------------------------------
# attention core, dense layers, loss
# ----------------------------------------------------------------------------------------------------------------------
class _FlashFn(Function):
    @staticmethod
    def forward(ctx, qkv):
        qkv = qkv.contiguous()
        lib = _capi.lib()
        B, C3, H, W = (int(v) for v in qkv.shape)
        Cc, L = C3 // 3, H * W
        o = torch.empty(B, Cc, H, W, device=qkv.device)
        lse = torch.empty(B, NUM_HEADS, L, device=qkv.device)
        need = C.c_int64(0)
        _capi.check(lib.hdiff_mha_flash_fwd_workspace(B, Cc, NUM_HEADS, L, C.byref(need)), "mha_flash_fwd_workspace")
        # scratch of the pre-split (bf16x3) forward: only allocated when that mode is on; freed when this call returns
        ws = torch.empty(need.value // 4 + 1, device=qkv.device) if need.value > 0 and lib.hdiff_get_contraction_mode() == 1 else None
        _capi.check(lib.hdiff_mha_flash_fwd_ws(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Cc, NUM_HEADS, L,
                                               None if ws is None else ws.data_ptr(), 0 if ws is None else need.value,
                                               _stream(qkv.device)), "mha_flash_fwd")
        ctx.save_for_backward(qkv, o, lse)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o, lse = ctx.saved_tensors
        d_o = d_o.contiguous()
        lib = _capi.lib()
        B, C3, H, W = (int(v) for v in qkv.shape)
        Cc, L = C3 // 3, H * W
        delta = torch.empty(B, NUM_HEADS, L, device=qkv.device)
        dqkv = torch.empty_like(qkv)
        need = C.c_int64(0)
        _capi.check(lib.hdiff_mha_flash_bwd_workspace(B, Cc, NUM_HEADS, L, C.byref(need)), "mha_flash_bwd_workspace")
        ws = torch.empty(need.value, dtype=torch.float32, device=qkv.device) if need.value else None   # dQ partial slabs
        _capi.check(lib.hdiff_mha_flash_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                            dqkv.data_ptr(), _p(ws), B, Cc, NUM_HEADS, L, _stream(qkv.device)),
                    "mha_flash_bwd")
        return dqkv


class _LinearFn(Function):
    """y = bias + f(x_row) W^T with an optional row gather (nn.Embedding) and an optional Swish on the input."""

    @staticmethod
    def forward(ctx, x, idx, W, bias, swish_input: bool, pad_row: int = -1):
        x, W = x.contiguous(), W.contiguous()
        ctx.pad_row = pad_row
        lib = _capi.lib()
        B = int(idx.shape[0]) if idx is not None else int(x.shape[0])
        N, K = int(W.shape[0]), int(W.shape[1])
        y = torch.empty(B, N, device=W.device)
        n_rows = int(x.shape[0]) if idx is not None else 0
        _capi.check(lib.hdiff_linear_rows(x.data_ptr(), _p(idx), n_rows, W.data_ptr(), _p(bias), y.data_ptr(), B, K, N,
                                          int(swish_input), 0, _stream(W.device)), "linear_rows")
        ctx.save_for_backward(x, idx, W)
        ctx.swish, ctx.has_bias = swish_input, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, idx, W = ctx.saved_tensors
        dy = dy.contiguous()
        lib = _capi.lib()
        B, N, K = int(dy.shape[0]), int(W.shape[0]), int(W.shape[1])
        n_rows = int(x.shape[0]) if idx is not None else 0
        d_W = torch.empty_like(W)
        d_b = torch.empty(N, device=W.device) if ctx.has_bias else None
        d_x = None
        if ctx.needs_input_grad[0]:
            d_x = torch.zeros_like(x) if idx is not None else torch.empty_like(x)
        _capi.check(lib.hdiff_linear_rows_bwd(x.data_ptr(), _p(idx), n_rows, W.data_ptr(), dy.data_ptr(), _p(d_x),
                                              d_W.data_ptr(), _p(d_b), B, K, N, int(ctx.swish), 0, ctx.pad_row,
                                              _stream(W.device)), "linear_rows_bwd")
        return d_x, None, d_W, d_b, None, None


class _SqErrFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        _capi.check(_capi.lib().hdiff_sq_err(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream(a.device)),
                    "sq_err")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, dl):
        a, b = ctx.saved_tensors
        dl = dl.contiguous()
        da = torch.empty_like(a)
        _capi.check(_capi.lib().hdiff_sq_err_bwd(a.data_ptr(), b.data_ptr(), dl.data_ptr(), da.data_ptr(), a.numel(),
                                                 _stream(a.device)), "sq_err_bwd")
        return da, None


class _AddFn(Function):
    """a + b for two small [B, C] vectors (h += temb_proj(...); h += cond_proj(...), ModelCondition.py:198-200)."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        _capi.check(_capi.lib().hdiff_axpby(C.c_float(1.0), a.contiguous().data_ptr(), C.c_float(1.0),
                                            b.contiguous().data_ptr(), out.data_ptr(), a.numel(), _stream(a.device)), "axpby")
        return out

    @staticmethod
    def backward(ctx, d):
        return d, d


class _GNAffineFn(Function):
    """nn.GroupNorm(32, C) with no activation behind it -- AttnBlock's pre-norm (ModelCondition.py:103)."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = x.contiguous()
        lib = _capi.lib()
        B, Cc, H, W = (int(v) for v in x.shape)
        scale, shift, mean, rstd = _gn_stats(x, None, gamma, beta, B, H * W)
        y = torch.empty_like(x)
        _capi.check(lib.hdiff_gn_affine_apply(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), B, Cc, H * W,
                                              _stream(x.device)), "gn_affine_apply")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        lib = _capi.lib()
        B, Cc, H, W = (int(v) for v in x.shape)
        dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(beta)
        ws = torch.empty(2 * B * Cc + 2 * B * GN_GROUPS, device=x.device)
        _capi.check(lib.hdiff_gn_affine_bwd(x.data_ptr(), Cc, B, H * W, GN_GROUPS, dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                            gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                            db.data_ptr(), _stream(x.device)), "gn_affine_bwd")
        return dx, dg, db


class _SingleHeadFn(Function):
    """softmax(q k^T * C^-1/2) v with ONE head of width C on stacked [q | k | v] rows -- the core of AttnBlock
    (ModelCondition.py:109-116).  Forward: the flash kernel up to 64 channels, the row kernel beyond; backward: the row / column
    passes of hdiff_mha_wide_bwd at any width (probabilities recomputed)."""

    @staticmethod
    def forward(ctx, qkv):
        qkv = qkv.contiguous()
        lib = _capi.lib()
        B, C3, H, W = (int(v) for v in qkv.shape)
        Cc, L = C3 // 3, H * W
        o = torch.empty(B, Cc, H, W, device=qkv.device)
        if Cc <= 64:
            _capi.check(lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), None, B, Cc, 1, L, _stream(qkv.device)), "mha_flash_fwd")
        else:
            _capi.check(lib.hdiff_mha_wide_fwd(qkv.data_ptr(), o.data_ptr(), B, Cc, L, _stream(qkv.device)), "mha_wide_fwd")
        ctx.save_for_backward(qkv)
        return o

    @staticmethod
    def backward(ctx, d_o):
        (qkv,) = ctx.saved_tensors
        d_o = d_o.contiguous()
        lib = _capi.lib()
        B, C3, H, W = (int(v) for v in qkv.shape)
        Cc, L = C3 // 3, H * W
        dqkv = torch.empty_like(qkv)
        ws = torch.empty(2 * B * L, device=qkv.device)
        _capi.check(lib.hdiff_mha_wide_bwd(qkv.data_ptr(), d_o.data_ptr(), dqkv.data_ptr(), ws.data_ptr(), B, Cc, L,
                                           _stream(qkv.device)), "mha_wide_bwd")
        return dqkv


# ----------------------------------------------------------------------------------------------------------------------
# public entry points
# ----------------------------------------------------------------------------------------------------------------------
def fused_conv(x0, x1, weight, bias, gn_w=None, gn_b=None, addvec=None, residual=None, k: int = 3, drop_p: float = 0.0):
    seed = int(torch.empty((), dtype=torch.int64).random_().item()) if drop_p > 0.0 else 0
    return _FusedConv.apply(x0, x1, weight, bias, gn_w, gn_b, addvec, residual, k, float(drop_p), seed)


def attn_block(ab, x):
    """AttnBlock.forward (ModelCondition.py:102-120) with gradients: GroupNorm -> q, k, v as one stacked 1x1 conv -> the
    single-head core -> 1x1 proj with the residual x in its epilogue."""
    hn = _GNAffineFn.apply(x, ab.group_norm.weight, ab.group_norm.bias)
    w_qkv = torch.cat([ab.proj_q.weight, ab.proj_k.weight, ab.proj_v.weight], dim=0)
    b_qkv = torch.cat([ab.proj_q.bias, ab.proj_k.bias, ab.proj_v.bias], dim=0)
    qkv = fused_conv(hn, None, w_qkv, b_qkv, k=1)
    o = _SingleHeadFn.apply(qkv)
    return fused_conv(o, None, ab.proj.weight, ab.proj.bias, residual=x, k=1)


def embed_mlp(seq, idx):
    """nn.Sequential(Embedding, Linear, Swish, Linear) of TimeEmbedding / ConditionalEmbedding."""
    pad = seq[0].padding_idx if seq[0].padding_idx is not None else -1
    h = _LinearFn.apply(seq[0].weight, idx, seq[1].weight, seq[1].bias, False, pad)
    return _LinearFn.apply(h, None, seq[3].weight, seq[3].bias, True)


def res_block(rb, xa, xb, temb, cemb, training: bool):
    """ResBlock.forward (ModelCondition.py:196-211) on the virtual concat [xa | xb]."""
    from torch import nn
    vec = _LinearFn.apply(temb, None, rb.temb_proj[1].weight, rb.temb_proj[1].bias, True)
    if cemb is not None:
        vec = _AddFn.apply(vec, _LinearFn.apply(cemb, None, rb.cond_proj[1].weight, rb.cond_proj[1].bias, True))
    h1 = fused_conv(xa, xb, rb.block1[2].weight, rb.block1[2].bias, rb.block1[0].weight, rb.block1[0].bias, addvec=vec, k=3)
    if isinstance(rb.shortcut, nn.Conv2d):
        sc = fused_conv(xa, xb, rb.shortcut.weight, rb.shortcut.bias, k=1)
    else:
        assert xb is None
        sc = xa
    p = rb.block2[2].p if (training and rb.block2[2].training) else 0.0
    h2 = fused_conv(h1, None, rb.block2[3].weight, rb.block2[3].bias, rb.block2[0].weight, rb.block2[0].bias, residual=sc, k=3,
                    drop_p=p)
    if isinstance(rb.attn, nn.MultiheadAttention):
        Cc = int(h2.shape[1])
        qkv = fused_conv(h2, None, rb.attn.in_proj_weight.view(3 * Cc, Cc, 1, 1), rb.attn.in_proj_bias, k=1)
        o = _FlashFn.apply(qkv)
        return fused_conv(o, None, rb.attn.out_proj.weight.view(Cc, Cc, 1, 1), rb.attn.out_proj.bias, k=1)
    return h2


def unet_forward_with_grad(model, x, t, labels):
    """UNet.forward (ModelCondition.py:255-276) with autograd through the HIP kernels."""
    from .DiffusionFreeGuidence.ModelCondition import DownSample, ResBlock, UpSample
    model.check_indices(t, labels)
    training = model.training
    temb = embed_mlp(model.time_embedding.timembedding, t)
    cemb = embed_mlp(model.cond_embedding.condEmbedding, labels)
    h = fused_conv(x, None, model.head.weight, model.head.bias, k=3)
    hs = [h]
    for layer in model.downblocks:
        if isinstance(layer, ResBlock):
            h = res_block(layer, h, None, temb, cemb, training)
        else:
            h = _DownFn.apply(h, layer.c1.weight, layer.c1.bias, layer.c2.weight, layer.c2.bias)
        hs.append(h)
    for layer in model.middleblocks:
        h = res_block(layer, h, None, temb, cemb, training)
    for layer in model.upblocks:
        if isinstance(layer, ResBlock):
            h = res_block(layer, h, hs.pop(), temb, cemb, training)
        else:
            u = _TConvFn.apply(h, layer.t.weight, layer.t.bias)
            h = fused_conv(u, None, layer.c.weight, layer.c.bias, k=3)
    assert len(hs) == 0
    return fused_conv(h, None, model.tail[2].weight, model.tail[2].bias, model.tail[0].weight, model.tail[0].bias, k=3)


def sq_err_with_grad(eps_hat, noise):
    return _SqErrFn.apply(eps_hat, noise)
