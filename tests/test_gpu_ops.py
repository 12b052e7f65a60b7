"""GPU parity, op level: every C-ABI kernel of libhdiff.so against the CPU oracle (oracle/cpu_path.py) on seeded inputs.

Tolerances (fp32 path; exact-fp32 MFMA, so only summation order differs from the reference):
  single op      <= 2e-5 * max|ref| (+1e-6)
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import hdiff_amd  # noqa: E402
from hdiff_amd import _capi, engine as E  # noqa: E402
from oracle import cpu_path as O  # noqa: E402

DEV = "cuda:0"


def close(got, ref, rel=2e-5, abs_=1e-6, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    tol = rel * ref.abs().max().item() + abs_
    assert err <= tol, f"{what}: max err {err:.3e} > tol {tol:.3e} (ref max {ref.abs().max().item():.3e})"


def run_conv(x0, x1, w, b, k, pad, stride=1, gn=None, addvec=None, residual=None):
    plan = E.Plan(DEV)
    B, C0, H, W = x0.shape
    pk = E._std_pack(plan, w.to(DEV), k, pad)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = plan.buf(B, w.shape[0], OH, OW)
    dg = lambda t: None if t is None else t.to(DEV).contiguous()
    plan.conv(dg(x0), dg(x1), pk, dg(b), out, B=B, H=H, W=W, VH=OH, VW=OW, in_stride=stride,
              gn=None if gn is None else (dg(gn[0]), dg(gn[1])), addvec=dg(addvec), residual=dg(residual))
    plan.pack_weights()
    plan.run()
    torch.cuda.synchronize()
    return out.clone()


@pytest.mark.parametrize("cin,cout,H,W,B", [(3, 32, 16, 16, 2), (32, 3, 16, 16, 2), (128, 256, 8, 8, 1), (32, 64, 40, 24, 1),
                                            (64, 64, 64, 64, 1), (128, 128, 32, 32, 2), (40, 72, 5, 7, 3)])
def test_conv3x3_plain(cin, cout, H, W, B):
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    close(run_conv(x, None, w, b, 3, 1), F.conv2d(x, w, b, padding=1), what="conv3x3")


def test_conv1x1_and_5x5_stride2():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 20, 12, generator=g)
    w = torch.randn(96, 64, 1, 1, generator=g) / 8
    b = torch.randn(96, generator=g)
    close(run_conv(x, None, w, b, 1, 0), F.conv2d(x, w, b), what="conv1x1")
    w5 = torch.randn(64, 64, 5, 5, generator=g) / 40
    close(run_conv(x, None, w5, b[:64], 5, 2, stride=2), F.conv2d(x, w5, b[:64], stride=2, padding=2), what="conv5x5s2")
    w3 = torch.randn(64, 64, 3, 3, generator=g) / 24
    close(run_conv(x, None, w3, None, 3, 1, stride=2), F.conv2d(x, w3, None, stride=2, padding=1), what="conv3x3s2")


@pytest.mark.parametrize("C0,C1,cout", [(64, 0, 128), (64, 31, 96), (128, 128, 40), (256, 0, 768)])
def test_conv1x1_direct_gemm_path(C0, C1, cout):
    """Full-resolution 1x1 convs (B*H*W >= 32768, H*W % 128 == 0) take the LDS-free GEMM kernel (conv1x1_direct.hip):
    two-pointer concat input, odd channel counts, channel tails, bias + per-sample vector + residual epilogue."""
    g = torch.Generator().manual_seed(C0 + C1 + cout)
    B, H, W = 2, 128, 128
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    cin = C0 + C1
    w = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    vec = torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    xin = x0 if x1 is None else torch.cat([x0, x1], dim=1)
    want = F.conv2d(xin.double(), w.double(), b.double()) + vec.double()[:, :, None, None] + res.double()
    close(run_conv(x0, x1, w, b, 1, 0, addvec=vec, residual=res), want.float(), rel=1e-5, what="conv1x1 direct")
    close(run_conv(x0, x1, w, None, 1, 0), F.conv2d(xin.double(), w.double()).float(), rel=1e-5, what="conv1x1 direct, no epilogue")


@pytest.mark.parametrize("C0,C1,cout,S,B", [(128, 0, 384, 128, 2), (128, 128, 40, 128, 2), (256, 0, 256, 64, 8), (64, 48, 128, 128, 2)])
def test_conv1x1_split_bf16_is_fp32_class(C0, C1, cout, S, B, bf16x3_mode):
    """conv1x1_x3.hip: the full-resolution 1x1 convs (attention projections, shortcuts) on bf16 triples in the bf16x3 mode --
    operands split in registers, no LDS.  Error against float64 in the class of the fp32-MFMA kernel's (conv1x1_direct.hip);
    concat input with the seam on a 16-channel boundary, a channel tail (cout 40), bias + vector + residual epilogue."""
    lib = bf16x3_mode
    g = torch.Generator().manual_seed(C0 + 3 * C1 + cout)
    cin = C0 + C1
    x0 = torch.randn(B, C0, S, S, generator=g) * torch.exp(torch.randn(1, C0, 1, 1, generator=g))
    x1 = torch.randn(B, C1, S, S, generator=g) if C1 else None
    w = torch.randn(cout, cin, 1, 1, generator=g) / math.sqrt(cin)
    b, vec, res = torch.randn(cout, generator=g), torch.randn(B, cout, generator=g), torch.randn(B, cout, S, S, generator=g)
    xin = (x0 if x1 is None else torch.cat([x0, x1], dim=1)).double()
    want = F.conv2d(xin, w.double(), b.double()) + vec.double()[:, :, None, None] + res.double()
    got_x3 = run_conv(x0, x1, w, b, 1, 0, addvec=vec, residual=res)
    _capi.check(lib.hdiff_set_contraction_mode(0))
    got_f32 = run_conv(x0, x1, w, b, 1, 0, addvec=vec, residual=res)
    _capi.check(lib.hdiff_set_contraction_mode(1))
    assert not torch.equal(got_x3, got_f32), "the split-bf16 1x1 kernel did not run"
    close(got_x3, want.float(), rel=1e-5, what="conv1x1 split-bf16")
    rms = lambda t: (t.double().cpu() - want).pow(2).mean().sqrt().item()
    worst = lambda t: (t.double().cpu() - want).abs().max().item()
    print(f"conv1x1 {cin}->{cout} at {S}: rms vs float64: triples {rms(got_x3):.3e}, fp32-MFMA {rms(got_f32):.3e}")
    assert rms(got_x3) <= 1.5 * rms(got_f32) + 1e-12, (rms(got_x3), rms(got_f32))
    assert worst(got_x3) <= 2.0 * worst(got_f32) + 1e-12, (worst(got_x3), worst(got_f32))
    close(run_conv(x0, x1, w, None, 1, 0), F.conv2d(xin, w.double()).float(), rel=1e-5, what="conv1x1 split-bf16, no epilogue")


def test_conv_fused_prologue_epilogue_concat():
    g = torch.Generator().manual_seed(9)
    B, C0, C1, Cout, H, W = 2, 64, 32, 64, 16, 16
    xa, xb = torch.randn(B, C0, H, W, generator=g), torch.randn(B, C1, H, W, generator=g)
    x = torch.cat([xa, xb], 1)
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 30
    b = torch.randn(Cout, generator=g)
    scale, shift = torch.rand(B, C0 + C1, generator=g) + 0.5, torch.randn(B, C0 + C1, generator=g)
    addvec, res = torch.randn(B, Cout, generator=g), torch.randn(B, Cout, H, W, generator=g)
    act = O.swish(x * scale[:, :, None, None] + shift[:, :, None, None])
    ref = F.conv2d(act, w, b, padding=1) + addvec[:, :, None, None] + res
    got = run_conv(xa, xb, w, b, 3, 1, gn=(scale, shift), addvec=addvec, residual=res)
    close(got, ref, what="conv fused")


@pytest.mark.parametrize("C0,C1,H,W,B", [(32, 0, 16, 16, 2), (256, 128, 8, 8, 2), (64, 0, 64, 64, 1), (96, 0, 6, 6, 2),
                                         (128, 0, 128, 128, 1)])
def test_groupnorm_scale_shift(C0, C1, H, W, B):
    g = torch.Generator().manual_seed(C0 + C1 + H)
    xa = torch.randn(B, C0, H, W, generator=g) * 2 + 0.7
    xb = torch.randn(B, C1, H, W, generator=g) - 0.3 if C1 else None
    Ct = C0 + C1
    gamma, beta = torch.randn(Ct, generator=g), torch.randn(Ct, generator=g)
    plan = E.Plan(DEV)
    dg = lambda t: None if t is None else t.to(DEV)
    sc, sh = plan.gn_scale_shift(dg(xa), dg(xb), dg(gamma), dg(beta), B, H * W)
    plan.run()
    x = xa if xb is None else torch.cat([xa, xb], 1)
    mean, rstd = O.group_norm_stats(x, 32, 1e-5)
    cpg = Ct // 32
    ref_sc = rstd.repeat_interleave(cpg, 1) * gamma[None]
    ref_sh = beta[None] - mean.repeat_interleave(cpg, 1) * ref_sc
    close(sc, ref_sc, what="gn scale")
    close(sh, ref_sh, rel=3e-5, what="gn shift")
    # and the full GN+Swish through the stand-alone apply kernel
    y = torch.empty(B, Ct, H, W, device=DEV)
    lib = _capi.lib()
    d_x = dg(x)
    _capi.check(lib.hdiff_gn_swish_apply(d_x.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), B, Ct, H * W,
                                         torch.cuda.current_stream().cuda_stream))
    close(y, O.swish(O.group_norm(x, 32, gamma, beta, 1e-5)), rel=3e-5, what="gn+swish")


@pytest.mark.parametrize("C0,C1,HW,B", [(128, 0, 65536, 2), (256, 128, 16384, 3), (64, 0, 4096, 1), (32, 0, 100, 4)])
def test_groupnorm_fused_finalize_equals_two_launches(C0, C1, HW, B):
    """hdiff_gn_scale_shift (one launch when a (sample, group) is one workgroup, else stats + merge inside the one call)
    against hdiff_gn_stats + hdiff_gn_finalize: the same bits."""
    g = torch.Generator(device=DEV).manual_seed(C0 + HW)
    xa = torch.randn(B, C0, HW, device=DEV, generator=g) * 3 - 1
    xb = torch.randn(B, C1, HW, device=DEV, generator=g) + 2 if C1 else None
    Ct = C0 + C1
    gamma, beta = torch.randn(Ct, device=DEV, generator=g), torch.randn(Ct, device=DEV, generator=g)
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    nsplit = max(1, min(32, HW // 4096))
    ws = torch.empty(B * 32 * nsplit * 3, device=DEV)
    sc0, sh0 = torch.empty(B, Ct, device=DEV), torch.empty(B, Ct, device=DEV)
    px = lambda t: None if t is None else t.data_ptr()
    _capi.check(lib.hdiff_gn_stats(px(xa), px(xb), C0, C1, B, HW, 32, nsplit, ws.data_ptr(), s))
    _capi.check(lib.hdiff_gn_finalize(ws.data_ptr(), B, Ct, 32, nsplit, gamma.data_ptr(), beta.data_ptr(), C.c_float(1e-5),
                                      sc0.data_ptr(), sh0.data_ptr(), None, None, s))
    for rep in range(2):
        ws2 = torch.full_like(ws, float("nan"))
        sc1, sh1 = torch.full_like(sc0, float("nan")), torch.full_like(sh0, float("nan"))
        _capi.check(lib.hdiff_gn_scale_shift(px(xa), px(xb), C0, C1, B, HW, 32, nsplit, ws2.data_ptr(), gamma.data_ptr(),
                                             beta.data_ptr(), C.c_float(1e-5), sc1.data_ptr(), sh1.data_ptr(), s))
        torch.cuda.synchronize()
        assert torch.equal(sc1, sc0) and torch.equal(sh1, sh0), rep


def test_linear_rows_multi_equals_the_launches_it_replaces():
    """All per-block projections of temb / cemb in one launch (ModelCondition.py:199-200) == per block
    linear_rows(temb) then linear_rows(cemb, accumulate): the same bits; jobs without a second term as well."""
    g = torch.Generator(device=DEV).manual_seed(11)
    B, K = 5, 512
    temb, cemb = torch.randn(B, K, device=DEV, generator=g), torch.randn(B, K, device=DEV, generator=g)
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    widths = [128, 128, 256, 40, 256, 3]
    ws_ = [(torch.randn(n, K, device=DEV, generator=g) / 20, torch.randn(n, device=DEV, generator=g),
            torch.randn(n, K, device=DEV, generator=g) / 20, torch.randn(n, device=DEV, generator=g)) for n in widths]
    for with_second in (True, False):
        want = []
        for n, (w0, b0, w1, b1) in zip(widths, ws_):
            y = torch.empty(B, n, device=DEV)
            _capi.check(lib.hdiff_linear_rows(temb.data_ptr(), None, 0, w0.data_ptr(), b0.data_ptr(), y.data_ptr(), B, K, n, 1, 0, s))
            if with_second:
                _capi.check(lib.hdiff_linear_rows(cemb.data_ptr(), None, 0, w1.data_ptr(), b1.data_ptr(), y.data_ptr(), B, K, n, 1, 1, s))
            want.append(y)
        outs = [torch.full((B, n), float("nan"), device=DEV) for n in widths]
        jobs = (_capi.LinearJob * len(widths))()
        first = 0
        for j, (n, (w0, b0, w1, b1)) in enumerate(zip(widths, ws_)):
            jobs[j].w0, jobs[j].b0, jobs[j].y, jobs[j].n, jobs[j].first = w0.data_ptr(), b0.data_ptr(), outs[j].data_ptr(), n, first
            jobs[j].w1, jobs[j].b1 = (w1.data_ptr(), b1.data_ptr()) if with_second else (None, None)
            first += n
        table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(DEV)
        _capi.check(lib.hdiff_linear_rows_multi(temb.data_ptr(), cemb.data_ptr() if with_second else None, table.data_ptr(),
                                                len(widths), first, B, K, s))
        torch.cuda.synchronize()
        for got, ref in zip(outs, want):
            assert torch.equal(got, ref)


def attention_core_ref(qkv, heads):
    B, C3, L = qkv.shape
    Cc = C3 // 3
    d = Cc // heads
    q, k, v = [z.reshape(B, heads, d, L).transpose(2, 3).double() for z in qkv.split(Cc, dim=1)]
    w = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(d), dim=-1)
    return (w @ v).transpose(2, 3).reshape(B, Cc, L).float()


@pytest.mark.parametrize("d,L,B", [(4, 64, 2), (8, 256, 1), (16, 64, 2), (16, 1024, 1), (32, 36, 2), (32, 576, 1),
                                   (16, 4096, 1), (32, 1000, 1), (8, 37, 1), (64, 256, 1), (64, 77, 2),
                                   (12, 1024, 1), (12, 50, 2), (24, 1024, 1), (24, 333, 1), (48, 512, 1), (48, 91, 2)])
def test_flash_attention_core(d, L, B):
    g = torch.Generator().manual_seed(d * 7 + L)
    heads = 8
    Cc = heads * d
    qkv = torch.randn(B, 3 * Cc, L, generator=g) * 1.5
    o = torch.empty(B, Cc, L, device=DEV)
    lib = _capi.lib()
    d_qkv = qkv.to(DEV)
    _capi.check(lib.hdiff_mha_flash_fwd(d_qkv.data_ptr(), o.data_ptr(), None, B, Cc, heads, L,
                                        torch.cuda.current_stream().cuda_stream), "mha")
    torch.cuda.synchronize()
    close(o, attention_core_ref(qkv, heads), rel=2e-5, abs_=2e-6, what=f"flash d={d} L={L}")


def test_flash_attention_online_softmax_rescale():
    """Force the running-max rescale: one key per later tile dominates (guide rule: test the rare branch)."""
    g = torch.Generator().manual_seed(3)
    heads, d, L, B = 8, 16, 512, 1
    Cc = heads * d
    qkv = torch.randn(B, 3 * Cc, L, generator=g)
    qkv[:, Cc:2 * Cc, 70] *= 6.0      # spike keys in tile 1, 4 and 7
    qkv[:, Cc:2 * Cc, 300] *= 12.0
    qkv[:, Cc:2 * Cc, 500] *= 20.0
    o = torch.empty(B, Cc, L, device=DEV)
    d_qkv = qkv.to(DEV)
    _capi.check(_capi.lib().hdiff_mha_flash_fwd(d_qkv.data_ptr(), o.data_ptr(), None, B, Cc, heads, L,
                                                torch.cuda.current_stream().cuda_stream), "mha")
    close(o, attention_core_ref(qkv, heads), rel=3e-5, abs_=3e-6, what="flash rescale")


@pytest.mark.parametrize("d", [16, 32])
def test_flash_attention_fixed_reference_overflow_falls_back(d):
    """The fast kernel keeps the first tile's max as its only softmax reference; a later key that beats it by > 2^7 (log2
    units) overflows exp2 on purpose -> the affected query blocks are poisoned and recomputed by the overflow-proof
    kernel.  Spike keys hard enough to force that, in some heads / query blocks only, and compare everything."""
    g = torch.Generator().manual_seed(8)
    heads, L, B = 8, 2048, 2
    Cc = heads * d
    qkv = torch.randn(B, 3 * Cc, L, generator=g)
    qkv[0, Cc + 2 * d:Cc + 3 * d, 700] *= 90.0          # head 2 of sample 0: key 700 (tile 10) dominates by hundreds of bits
    qkv[1, Cc + 5 * d:Cc + 6 * d, 1999] *= 150.0        # head 5 of sample 1: the very last tile
    qkv[1, 0 * d:1 * d, 300:364] *= 40.0                # huge queries in one 64-query stretch of head 0
    o = torch.empty(B, Cc, L, device=DEV)
    lse = torch.empty(B, heads, L, device=DEV)
    d_qkv = qkv.to(DEV)
    _capi.check(_capi.lib().hdiff_mha_flash_fwd(d_qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Cc, heads, L,
                                                torch.cuda.current_stream().cuda_stream), "mha")
    ref = attention_core_ref(qkv, heads)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    close(o, ref, rel=3e-5, abs_=3e-6, what="flash overflow fallback")
    # log-sum-exp (log2 domain) against fp64
    q, k, _ = [z.reshape(B, heads, d, L).transpose(2, 3).double() for z in qkv.split(Cc, dim=1)]
    s2 = (q @ k.transpose(2, 3)) / math.sqrt(d) * math.log2(math.e)
    ref_lse = torch.logsumexp(s2 * math.log(2.0), dim=-1) / math.log(2.0)
    close(lse, ref_lse.float(), rel=1e-5, abs_=1e-4, what="lse2")


@pytest.fixture
def bf16x3_mode():
    lib = _capi.lib()
    before = lib.hdiff_get_contraction_mode()
    _capi.check(lib.hdiff_set_contraction_mode(1))
    yield lib
    _capi.check(lib.hdiff_set_contraction_mode(before))


def _flash(lib, qkv, heads, want_lse=False, workspace=False):
    """workspace=True: hdiff_mha_flash_fwd_ws with the scratch the library asks for (the pre-split kernel in bf16x3 mode)."""
    B, C3, L = qkv.shape
    Cc = C3 // 3
    o = torch.empty(B, Cc, L, device=DEV)
    lse = torch.empty(B, heads, L, device=DEV) if want_lse else None
    d_qkv = qkv.to(DEV)
    s = torch.cuda.current_stream().cuda_stream
    if workspace:
        need = C.c_int64(-1)
        _capi.check(lib.hdiff_mha_flash_fwd_workspace(B, Cc, heads, L, C.byref(need)), "mha ws query")
        ws = torch.empty(need.value // 4 + 1, device=DEV) if need.value > 0 else None
        _capi.check(lib.hdiff_mha_flash_fwd_ws(d_qkv.data_ptr(), o.data_ptr(), lse.data_ptr() if want_lse else None, B, Cc,
                                               heads, L, None if ws is None else ws.data_ptr(), max(need.value, 0), s), "mha ws")
    else:
        _capi.check(lib.hdiff_mha_flash_fwd(d_qkv.data_ptr(), o.data_ptr(), lse.data_ptr() if want_lse else None, B, Cc, heads,
                                            L, s), "mha")
    torch.cuda.synchronize()
    return o, lse


@pytest.mark.parametrize("workspace", [False, True], ids=["split-in-loop", "pre-split"])
@pytest.mark.parametrize("d,L,B,scale", [(16, 1024, 2, 1.0), (16, 4096, 1, 3.0), (32, 2048, 1, 1.0), (32, 512, 2, 2.0)])
def test_flash_attention_split_bf16_is_fp32_class(d, L, B, scale, workspace, bf16x3_mode):
    """HDIFF_CONTRACT_BF16X3, the split-operand mode: fp32 operands as 16-bit pieces on the 16-bit MFMA -- with a workspace the fp16-pair
    kernels (attention_h2.hip at d_head 16, attention_x3p.hip at 32), without one the in-loop bf16-triple kernel (attention_x3.hip).
    Claim checked here: the error against float64 is of the same class as the fp32-MFMA kernel's (not merely inside the tolerance)."""
    lib = bf16x3_mode
    g = torch.Generator().manual_seed(100 + d + L)
    heads = 8
    qkv = torch.randn(B, 3 * heads * d, L, generator=g) * scale
    ref = attention_core_ref(qkv, heads).double()
    o_x3, _ = _flash(lib, qkv, heads, workspace=workspace)
    if workspace:
        # with a workspace other programs run: d_head 32 the 32x32x16 kernel on pre-split bf16 triples, d_head 16 the
        # fp16-pair P.V kernel (attention_h2.hip) -- other pieces, other summation order, so other bits
        assert not torch.equal(o_x3, _flash(lib, qkv, heads)[0]), "the workspace path did not run its own kernel"
    assert lib.hdiff_get_contraction_mode() == 1
    _capi.check(lib.hdiff_set_contraction_mode(0))
    o_f32, _ = _flash(lib, qkv, heads)
    _capi.check(lib.hdiff_set_contraction_mode(1))
    assert not torch.equal(o_x3, o_f32), "the split-bf16 kernel did not run"
    close(o_x3, ref.float(), rel=2e-5, abs_=2e-6, what=f"split-bf16 d={d} L={L}")
    rms = lambda t: (t.double().cpu() - ref).pow(2).mean().sqrt().item()
    worst = lambda t: (t.double().cpu() - ref).abs().max().item()
    assert rms(o_x3) <= 1.25 * rms(o_f32) + 1e-12, (rms(o_x3), rms(o_f32))
    assert worst(o_x3) <= 2.0 * worst(o_f32) + 1e-12, (worst(o_x3), worst(o_f32))


def _h2_case(name, d, L, g):
    """Inputs that stress what the fp16-pair P.V kernel (attention_h2.hip) adds to the bf16-triple one: a softmax reference
    that has to move, and fp16's range."""
    heads = 8
    qkv = torch.randn(1, 3 * heads * d, L, generator=g)
    Cc = heads * d
    if name == "ramp":            # the row maximum keeps rising along the keys
        qkv[:, Cc:2 * Cc] *= torch.linspace(0.3, 5.0, L)
    elif name == "peaked":        # scores with a standard deviation of 13 in the exp2 domain
        qkv *= 3.0
    elif name == "late-spikes":   # a few keys far above everything before them (jumps of ~2^100 in P), in the middle and at the end
        qkv[:, Cc:2 * Cc, L // 2 + 5] *= 25.0
        qkv[:, Cc:2 * Cc, L - 3] *= 40.0
    elif name == "wide-v":        # V channels spanning 2^30 between and 2^12 within rows
        qkv[:, 2 * Cc:] *= torch.exp(torch.randn(1, Cc, 1, generator=g) * 6) * torch.exp(torch.randn(1, Cc, L, generator=g) * 2)
    elif name == "tiny-v":        # a head whose V is denormal-small next to normal ones
        qkv[:, 2 * Cc:2 * Cc + d] *= 1e-30
    elif name == "quiet-neighbour":   # every other block of 32 queries has scores 2^-20 ... 2^20 times its neighbours': queries that
        blk = (torch.arange(L) // 32) % 2 == 0                      # share a lane (d_head 32) must not share a softmax reference
        qkv[:, :Cc, blk] *= 6.0
        qkv[:, :Cc, ~blk] *= 0.05
    return qkv


@pytest.mark.parametrize("d", [16, 32])
@pytest.mark.parametrize("name", ["ramp", "peaked", "late-spikes", "wide-v", "tiny-v", "quiet-neighbour"])
def test_flash_attention_fp16_pairs_moving_reference_and_ranges(name, d, bf16x3_mode):
    """attention_h2.hip (d_head 16) / attention_x3p.hip PVH (d_head 32), with a workspace: P as two fp16 pieces needs a
    reference that MOVES (fp16 ends at 65504), one per query, and V scaled per channel row.  Error against float64 in the class
    of the fp32-MFMA kernel's on the same inputs, and -- in a process that skips the check pass -- not a single NaN: the
    kernel handled every row itself instead of poisoning it for the fp32 kernel behind it."""
    lib = bf16x3_mode
    L, heads = 4096, 8
    g = torch.Generator().manual_seed({"ramp": 1, "peaked": 2, "late-spikes": 3, "wide-v": 4, "tiny-v": 5, "quiet-neighbour": 6}[name])
    qkv = _h2_case(name, d, L, g)
    assert not torch.equal(_flash(lib, qkv, heads, workspace=True)[0], _flash(lib, qkv, heads)[0]), "the workspace path did not run its own kernel"
    ref = attention_core_ref(qkv, heads).double()
    o_h2, lse = _flash(lib, qkv, heads, want_lse=True, workspace=True)
    _capi.check(lib.hdiff_set_contraction_mode(0))
    o_f32, _ = _flash(lib, qkv, heads)
    _capi.check(lib.hdiff_set_contraction_mode(1))
    assert torch.isfinite(o_h2).all()
    # the log-sum-exp the backward reads: log2(l) plus the reference the row ENDED with, however often it moved
    Cc = heads * d
    q, k, _ = [z.reshape(1, heads, d, L).transpose(2, 3).double() for z in qkv.split(Cc, dim=1)]
    s2 = (q @ k.transpose(2, 3)) / math.sqrt(d) * math.log2(math.e)
    close(lse, (torch.logsumexp(s2 * math.log(2.0), dim=-1) / math.log(2.0)).float(), rel=1e-5, abs_=1e-4, what=f"lse2 {name} d={d}")
    scale = ref.abs().amax(dim=2, keepdim=True).clamp_min(1e-300)          # per channel row: rows differ by 2^30 in "wide-v"
    rms = lambda t: ((t.double().cpu() - ref) / scale).pow(2).mean().sqrt().item()
    worst = lambda t: ((t.double().cpu() - ref) / scale).abs().max().item()
    print(f"h2 {name}: rms {rms(o_h2):.3e} (fp32-MFMA kernel {rms(o_f32):.3e}), worst {worst(o_h2):.3e} ({worst(o_f32):.3e})")
    assert rms(o_h2) <= 1.25 * rms(o_f32) + 1e-12, (rms(o_h2), rms(o_f32))
    assert worst(o_h2) <= 2.0 * worst(o_f32) + 1e-12, (worst(o_h2), worst(o_f32))


def test_flash_attention_fp16_pairs_keeps_every_row_in_the_kernel():
    """The same inputs with the check pass switched off (HDIFF_NO_CHECK_PASS, read once per process): no NaN anywhere, i.e.
    the moving reference kept every P inside fp16 -- none of these rows was handed to the fp32 kernel."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, hdiff_amd
from hdiff_amd import _capi
import test_gpu_ops as T
lib = hdiff_amd.lib(); hdiff_amd.set_contraction_mode("bf16x3")
for d in (16, 32):
    for i, name in enumerate(["ramp", "peaked", "late-spikes", "wide-v"]):
        qkv = T._h2_case(name, d, 4096, torch.Generator().manual_seed(i + 1))
        o, lse = T._flash(lib, qkv, 8, want_lse=True, workspace=True)
        assert torch.isfinite(o).all() and torch.isfinite(lse).all(), (name, d)
        ref = T.attention_core_ref(qkv, 8)
        assert ((o.cpu() - ref).abs().amax(dim=2) <= 3e-5 * ref.abs().amax(dim=2) + 1e-30).all(), (name, d)
print("H2_ROWS_OK")
''' % (root, os.path.join(root, "tests"))
    env = dict(os.environ, HDIFF_NO_CHECK_PASS="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "H2_ROWS_OK" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]


@pytest.mark.parametrize("workspace", [False, True], ids=["split-in-loop", "pre-split"])
def test_flash_attention_split_bf16_overflow_falls_back(workspace, bf16x3_mode):
    """Same fixed-reference protocol as the fp32 fast kernel: spiked keys / queries overflow exp2 on purpose; the poisoned
    query blocks are recomputed by the overflow-proof kernel."""
    lib = bf16x3_mode
    g = torch.Generator().manual_seed(9)
    heads, d, L, B = 8, 16, 2048, 2
    Cc = heads * d
    qkv = torch.randn(B, 3 * Cc, L, generator=g)
    qkv[0, Cc + 2 * d:Cc + 3 * d, 700] *= 90.0
    qkv[1, Cc + 5 * d:Cc + 6 * d, 1999] *= 150.0
    qkv[1, 0 * d:1 * d, 300:364] *= 40.0
    o, lse = _flash(lib, qkv, heads, want_lse=True, workspace=workspace)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    close(o, attention_core_ref(qkv, heads), rel=3e-5, abs_=3e-6, what="split-bf16 overflow fallback")
    q, k, _ = [z.reshape(B, heads, d, L).transpose(2, 3).double() for z in qkv.split(Cc, dim=1)]
    s2 = (q @ k.transpose(2, 3)) / math.sqrt(d) * math.log2(math.e)
    ref_lse = torch.logsumexp(s2 * math.log(2.0), dim=-1) / math.log(2.0)
    close(lse, ref_lse.float(), rel=1e-5, abs_=1e-4, what="lse2 (split-bf16)")


@pytest.mark.parametrize("C0,C1,cout,H,W,B,use_gn", [(64, 0, 64, 64, 64, 16, True), (128, 64, 128, 64, 96, 4, True),
                                                        (32, 16, 96, 40, 72, 8, False), (256, 0, 256, 32, 32, 12, True)])
def test_conv3x3_split_bf16_is_fp32_class(C0, C1, cout, H, W, B, use_gn, bf16x3_mode):
    """HDIFF_CONTRACT_BF16X3 for the 3x3 convolutions (conv3x3_x3.hip): error against float64 of the same class as the
    fp32-MFMA kernel's, with the fused GroupNorm/Swish prologue, concat input, bias + vector + residual epilogue."""
    lib = bf16x3_mode
    g = torch.Generator().manual_seed(C0 * 7 + cout + H)
    cin = C0 + C1
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    vec = torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    gn = (torch.rand(B, cin, generator=g) + 0.5, torch.randn(B, cin, generator=g) * 0.3) if use_gn else None
    xin = (x0 if x1 is None else torch.cat([x0, x1], dim=1)).double()
    if use_gn:
        a = xin * gn[0].double()[:, :, None, None] + gn[1].double()[:, :, None, None]
        xin = a * torch.sigmoid(a)
    want = F.conv2d(xin, w.double(), b.double(), padding=1) + vec.double()[:, :, None, None] + res.double()
    got_x3 = run_conv(x0, x1, w, b, 3, 1, gn=gn, addvec=vec, residual=res)
    _capi.check(lib.hdiff_set_contraction_mode(0))
    got_f32 = run_conv(x0, x1, w, b, 3, 1, gn=gn, addvec=vec, residual=res)
    _capi.check(lib.hdiff_set_contraction_mode(1))
    assert not torch.equal(got_x3, got_f32), "the split-bf16 kernel did not run"
    close(got_x3, want.float(), rel=1e-5, what="conv3x3 split-bf16")
    rms = lambda t: (t.double().cpu() - want).pow(2).mean().sqrt().item()
    assert rms(got_x3) <= 1.5 * rms(got_f32) + 1e-12, (rms(got_x3), rms(got_f32))


def run_conv_gn(x0, x1, gamma, beta, w, b, addvec=None, residual=None, pairs=True):
    """GroupNorm statistics -> fused GroupNorm / Swish prologue -> 3x3 conv through the plan, as the U-Net emits it; with
    ``pairs`` the conv may take the fp16-pair kernel (bf16x3 mode), without it the plan forgets where the statistics came from
    and the bf16-triple kernel runs."""
    plan = E.Plan(DEV)
    B, C0, H, W = x0.shape
    dg = lambda t: None if t is None else t.to(DEV).contiguous()
    d_x0, d_x1, d_gamma, d_beta = dg(x0), dg(x1), dg(gamma), dg(beta)
    pk = E._std_pack(plan, w.to(DEV), 3, 1)
    out = plan.buf(B, w.shape[0], H, W)
    gn = plan.gn_scale_shift(d_x0, d_x1, d_gamma, d_beta, B, H * W)
    if not pairs:
        plan._gn_src.clear()
    plan.conv(d_x0, d_x1, pk, dg(b), out, B=B, H=H, W=W, VH=H, VW=W, gn=gn, addvec=dg(addvec), residual=dg(residual))
    assert (pk.wp2 is not None) == pairs
    plan.pack_weights()
    plan.run()
    torch.cuda.synchronize()
    return out.clone()


def _gn_swish_conv_f64(x0, x1, gamma, beta, w, b, addvec, residual):
    xin = (x0 if x1 is None else torch.cat([x0, x1], dim=1)).double()
    a = F.group_norm(xin, 32, gamma.double(), beta.double(), eps=1e-5)
    a = a * torch.sigmoid(a)
    want = F.conv2d(a, w.double(), None if b is None else b.double(), padding=1)
    if addvec is not None:
        want = want + addvec.double()[:, :, None, None]
    if residual is not None:
        want = want + residual.double()
    return want


@pytest.mark.parametrize("C0,C1,cout,H,W,B", [(64, 0, 64, 64, 64, 16), (128, 64, 128, 64, 96, 4), (256, 0, 256, 32, 32, 12),
                                               (128, 0, 128, 128, 128, 2)])
def test_conv3x3_fp16_pairs_is_fp32_class(C0, C1, cout, H, W, B, bf16x3_mode):
    """conv3x3_x3.hip, PAIR: the plain 3x3 conv behind GroupNorm + Swish with both operands as fp16 pairs staged at a power of
    two fixed by the GroupNorm weights (three products instead of six).  Error against float64 in the class of the fp32-MFMA
    kernel's (the bf16-triple kernel's gate) and not above the bf16-triple kernel's own; concat input, channels with very
    different scales and offsets, bias + vector + residual epilogue."""
    lib = bf16x3_mode
    g = torch.Generator().manual_seed(C0 * 5 + cout + H)
    cin = C0 + C1
    chan = lambda c: torch.exp(torch.randn(1, c, 1, 1, generator=g) * 1.5)
    x0 = torch.randn(B, C0, H, W, generator=g) * chan(C0) + torch.randn(1, C0, 1, 1, generator=g) * 3
    x1 = torch.randn(B, C1, H, W, generator=g) * chan(C1) if C1 else None
    gamma, beta = torch.rand(cin, generator=g) * 1.5 + 0.25, torch.randn(cin, generator=g) * 0.5
    w = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b, vec, res = torch.randn(cout, generator=g), torch.randn(B, cout, generator=g), torch.randn(B, cout, H, W, generator=g)
    want = _gn_swish_conv_f64(x0, x1, gamma, beta, w, b, vec, res)
    got_h2 = run_conv_gn(x0, x1, gamma, beta, w, b, vec, res)
    got_x3 = run_conv_gn(x0, x1, gamma, beta, w, b, vec, res, pairs=False)
    _capi.check(lib.hdiff_set_contraction_mode(0))
    got_f32 = run_conv_gn(x0, x1, gamma, beta, w, b, vec, res, pairs=False)
    _capi.check(lib.hdiff_set_contraction_mode(1))
    assert not torch.equal(got_h2, got_x3) and not torch.equal(got_h2, got_f32), "the fp16-pair kernel did not run"
    close(got_h2, want.float(), rel=1e-5, what="conv3x3 fp16 pairs")
    rms = lambda t: (t.double().cpu() - want).pow(2).mean().sqrt().item()
    worst = lambda t: (t.double().cpu() - want).abs().max().item()
    print(f"conv3x3 {cin}->{cout} {H}x{W}: rms vs float64: pairs {rms(got_h2):.3e}, triples {rms(got_x3):.3e}, fp32-MFMA {rms(got_f32):.3e}")
    assert rms(got_h2) <= 1.5 * rms(got_f32) + 1e-12, (rms(got_h2), rms(got_f32))
    assert rms(got_h2) <= 1.25 * rms(got_x3) + 1e-12, (rms(got_h2), rms(got_x3))
    assert worst(got_h2) <= 2.0 * worst(got_f32) + 1e-12, (worst(got_h2), worst(got_f32))


@pytest.mark.parametrize("name", ["lone-spike", "huge-gamma", "tiny-gamma", "huge-weights", "tiny-weights"])
def test_conv3x3_fp16_pairs_range(name, bf16x3_mode):
    """The fp16 range of the PAIR kernel: a group whose only non-zero element is normalised to sqrt(n - 1) -- the largest value
    GroupNorm can produce, the bound the staging power of two is made for --, GroupNorm weights and conv weights of extreme
    magnitude.  Finite, and as close to float64 as the fp32-MFMA kernel."""
    lib = bf16x3_mode
    g = torch.Generator().manual_seed(11)
    B, Cc, H, W = 16, 64, 64, 64          # 256 workgroups: large enough for the split-operand kernels
    x = torch.randn(B, Cc, H, W, generator=g)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    w = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9)
    if name == "lone-spike":
        x[:, :2] = 0.0
        x[:, 0, 17, 23] = 5.0          # group 0 = channels 0, 1: xhat there is sqrt(2 * 4096 - 1) = 90.5
        gamma[0], beta[0] = 2.0, 1.0
    elif name == "huge-gamma":
        gamma *= 3.0e4; beta *= 1.0e4
    elif name == "tiny-gamma":
        gamma *= 1.0e-6; beta *= 1.0e-6
    elif name == "huge-weights":
        w *= 1.0e12
    elif name == "tiny-weights":
        w *= 1.0e-12
    want = _gn_swish_conv_f64(x, None, gamma, beta, w, None, None, None)
    got = run_conv_gn(x, None, gamma, beta, w, None)
    _capi.check(lib.hdiff_set_contraction_mode(0))
    got_f32 = run_conv_gn(x, None, gamma, beta, w, None, pairs=False)
    _capi.check(lib.hdiff_set_contraction_mode(1))
    assert torch.isfinite(got).all()
    assert not torch.equal(got, got_f32), "the fp16-pair kernel did not run"
    scale = want.abs().max().item()
    rms = lambda t: (t.double().cpu() - want).pow(2).mean().sqrt().item() / scale
    worst = lambda t: (t.double().cpu() - want).abs().max().item() / scale
    print(f"{name}: rms {rms(got):.3e} (fp32-MFMA {rms(got_f32):.3e}), worst {worst(got):.3e} ({worst(got_f32):.3e})")
    assert rms(got) <= 1.5 * rms(got_f32) + 1e-12 and worst(got) <= 2.0 * worst(got_f32) + 1e-12


def test_conv3x3_fp16_pairs_outside_the_stated_range_is_loud(bf16x3_mode):
    """act_scale is the caller's statement about the staged activations' range; scale / shift that are not the statistics of
    x break it -- the answer is then NaN, never a silently wrong number."""
    g = torch.Generator().manual_seed(12)
    B, Cc, H, W = 16, 64, 64, 64
    x = torch.randn(B, Cc, H, W, generator=g)
    gamma, beta = torch.ones(Cc), torch.zeros(Cc)
    w = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9)
    d_x, d_gamma, d_beta = x.to(DEV), gamma.to(DEV), beta.to(DEV)
    pre = E.Plan(DEV)                       # the statistics in a plan of their own: the conv plan below is run twice
    gn = pre.gn_scale_shift(d_x, None, d_gamma, d_beta, B, H * W)
    pre.run()
    plan = E.Plan(DEV)
    plan._gn_src[id(gn[0])] = pre._gn_src[id(gn[0])]
    pk = E._std_pack(plan, w.to(DEV), 3, 1)
    out = plan.buf(B, Cc, H, W)
    plan.conv(d_x, None, pk, None, out, B=B, H=H, W=W, VH=H, VW=W, gn=gn)
    assert pk.wp2 is not None
    plan.pack_weights()
    plan.run()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    gn[0].mul_(1.0e4)                      # "statistics" of some other tensor: values up to 4e4 where the bound says 128
    plan.run()
    torch.cuda.synchronize()
    assert torch.isnan(out).any()


def test_gn_act_scale_and_weight_pairs_through_the_c_abi():
    """hdiff_gn_act_scale and hdiff_pack_conv_weight_h2 called straight through the C ABI: the staged range is the stated bound
    (sqrt(n - 1) max|gamma| + max|beta|) times gain, as a power of two with a factor two to spare; the packed pieces add up to
    w 2^t to 2^-22 of the largest weight, the tail holds 2^-t and 2^t, channel padding is zero."""
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(21)
    for Cc, n, gain in ((128, 4 * 65536, 1.0), (384, 12 * 4096, 1.0 / 0.85), (64, 2 * 64, 1.0)):
        gamma, beta = (torch.randn(Cc, generator=g) * 3).to(DEV), (torch.randn(Cc, generator=g) * 2).to(DEV)
        out = torch.zeros(2, device=DEV)
        _capi.check(lib.hdiff_gn_act_scale(gamma.data_ptr(), beta.data_ptr(), Cc, C.c_int64(n), C.c_float(gain), out.data_ptr(), s))
        bound = ((math.sqrt(n - 1) * gamma.abs() + beta.abs()).max().item()) * gain
        sc, inv = out.tolist()
        assert sc * inv == 1.0 and math.log2(sc) == round(math.log2(sc))              # an exact power of two and its inverse
        assert 2.0 ** 13 <= bound * sc * (1 + 1e-6) and bound * sc < 2.0 ** 14 * (1 + 1e-6), (bound, sc)
    cout, cin, cpad = 96, 48, 128
    w = torch.randn(cout, cin, 3, 3, generator=g) * torch.exp(torch.randn(cout, 1, 1, 1, generator=g) * 2)
    dw = w.to(DEV)
    words = C.c_int64(0)
    _capi.check(lib.hdiff_pack_conv_weight_h2_words(cout, cin, cpad, C.byref(words)))
    assert words.value == (cin // 16) * 9 * 2 * cpad * 8 + 4
    wp = torch.empty(words.value, dtype=torch.int32, device=DEV)
    _capi.check(lib.hdiff_pack_conv_weight_h2(dw.data_ptr(), wp.data_ptr(), cout, cin, cpad, s))
    torch.cuda.synchronize()
    tail = wp[-4:].view(torch.float32).tolist()
    inv_t, t = tail[1], tail[2]
    assert inv_t * t == 1.0 and 2.0 ** 14 <= w.abs().max().item() * t < 2.0 ** 15
    body = wp[:-4].view(torch.float16).float().reshape(cin // 16, 9, 2, cpad, 16).cpu()       # [chunk][tap][piece][co][16 ci]
    back = (body[:, :, 0] + body[:, :, 1]) * inv_t                                             # [chunk][tap][co][16 ci]
    want = w.reshape(cout, cin // 16, 16, 9).permute(1, 3, 0, 2)                               # [chunk][tap][co][16 ci]
    assert (back[:, :, :cout] - want).abs().max().item() <= 2.0 ** -22 * w.abs().max().item()
    assert (back[:, :, :cout] - want).abs().div(want.abs().clamp_min(2.0 ** -17 * w.abs().max().item())).max().item() <= 2.0 ** -22
    assert body[:, :, :, cout:].abs().max().item() == 0.0


def test_linear_rows_and_gather():
    g = torch.Generator().manual_seed(1)
    table = torch.randn(20, 128, generator=g)
    idx = torch.tensor([3, 0, 19, 7])
    W1, b1 = torch.randn(512, 128, generator=g) / 11, torch.randn(512, generator=g)
    y = torch.empty(4, 512, device=DEV)
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    # device copies are held in variables: a temporary's memory would be recycled before the kernel reads it
    d_table, d_idx, d_W1, d_b1 = table.to(DEV), idx.to(DEV), W1.to(DEV), b1.to(DEV)
    _capi.check(lib.hdiff_linear_rows(d_table.data_ptr(), d_idx.data_ptr(), 20, d_W1.data_ptr(), d_b1.data_ptr(),
                                      y.data_ptr(), 4, 128, 512, 0, 0, s))
    close(y, table[idx] @ W1.t() + b1, what="gather+linear")
    x = torch.randn(4, 512, generator=g)
    W2 = torch.randn(96, 512, generator=g) / 22
    y2 = torch.ones(4, 96, device=DEV)
    d_x, d_W2 = x.to(DEV), W2.to(DEV)
    _capi.check(lib.hdiff_linear_rows(d_x.data_ptr(), None, 0, d_W2.data_ptr(), None, y2.data_ptr(), 4, 512, 96, 1, 1, s))
    close(y2, 1.0 + O.swish(x) @ W2.t(), what="swish linear accumulate")


def test_ddpm_step_bit_exact_and_nan_flag():
    g = torch.Generator().manual_seed(2)
    n, T, w = 3 * 33 * 31, 50, 1.8
    x, ec, eu, z = [torch.randn(n, generator=g) for _ in range(4)]
    sched = O.sampler_schedule(1e-4, 0.028, T)
    var = O.sampler_variance_table(sched)
    c1, c2, sg = sched["coeff1"].float(), sched["coeff2"].float(), torch.sqrt(var.float())
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    d = lambda t: t.to(DEV)
    dx, dec, deu, dz, dc1, dc2, dsg = d(x), d(ec), d(eu), d(z), d(c1), d(c2), d(sg)
    for step in (37, 0):
        st = torch.tensor([step], dtype=torch.int32, device=DEV)
        out = torch.empty(n, device=DEV)
        _capi.check(lib.hdiff_ddpm_step(dx.data_ptr(), dec.data_ptr(), deu.data_ptr(), dz.data_ptr(), out.data_ptr(),
                                        dc1.data_ptr(), dc2.data_ptr(), dsg.data_ptr(), st.data_ptr(), T, C.c_double(w),
                                        C.c_uint64(0), flag.data_ptr(), n, s))
        eps = (1. + w) * ec - w * eu                          # DiffusionCondition.py:78
        mean = c1[step] * x - c2[step] * eps                  # :68-70
        ref = mean + sg[step] * z if step > 0 else mean       # :91-95
        assert torch.equal(out.cpu(), ref), f"ddpm_step not bit-exact at step {step}"
    assert flag.item() == 0
    dec[5] = float("nan")
    st = torch.tensor([3], dtype=torch.int32, device=DEV)
    _capi.check(lib.hdiff_ddpm_step(dx.data_ptr(), dec.data_ptr(), deu.data_ptr(), dz.data_ptr(), out.data_ptr(),
                                    dc1.data_ptr(), dc2.data_ptr(), dsg.data_ptr(), st.data_ptr(), T, C.c_double(w),
                                    C.c_uint64(0), flag.data_ptr(), n, s))
    assert flag.item() == 1
    # a step counter outside the schedule is clamped, never an out-of-bounds read of the tables
    dec[5] = 0.0
    for bad_step, as_step in ((-7, 0), (10 ** 6, T - 1)):
        st = torch.tensor([bad_step], dtype=torch.int32, device=DEV)
        _capi.check(lib.hdiff_ddpm_step(dx.data_ptr(), dec.data_ptr(), deu.data_ptr(), dz.data_ptr(), out.data_ptr(),
                                        dc1.data_ptr(), dc2.data_ptr(), dsg.data_ptr(), st.data_ptr(), T, C.c_double(w),
                                        C.c_uint64(0), flag.data_ptr(), n, s))
        eps = (1. + w) * dec.cpu() - w * eu
        mean = c1[as_step] * x - c2[as_step] * eps
        assert torch.equal(out.cpu(), mean + sg[as_step] * z if as_step > 0 else mean)


def test_ddpm_step_loop_bookkeeping():
    """hdiff_ddpm_step_loop = the same update + the loop's bookkeeping in one launch: x_next also lands in the two halves of
    the next UNet input, the device-resident step goes down by one and the time vector is refilled -- by the LAST workgroup
    to finish, replay after replay without a reset of its counter."""
    g = torch.Generator().manual_seed(3)
    B, per = 3, 3 * 40 * 36
    n, T, w = B * per, 20, 1.8
    x, ec, eu = [torch.randn(n, generator=g).to(DEV) for _ in range(3)]
    sched = O.sampler_schedule(1e-4, 0.028, T)
    c1, c2 = sched["coeff1"].float().to(DEV), sched["coeff2"].float().to(DEV)
    sg = torch.sqrt(O.sampler_variance_table(sched).float()).to(DEV)
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    step = torch.tensor([2], dtype=torch.int32, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    done = torch.zeros(1, dtype=torch.int32, device=DEV)
    t_next = torch.full((2 * B,), -5, dtype=torch.int64, device=DEV)
    z = torch.randn(n, generator=g).to(DEV)
    xin = torch.zeros(2 * n, device=DEV)
    cur = x.clone()
    d = _capi.DdpmLoopDesc()
    d.x, d.eps_c, d.eps_u, d.noise, d.x_next = cur.data_ptr(), ec.data_ptr(), eu.data_ptr(), z.data_ptr(), cur.data_ptr()
    d.coeff1, d.coeff2, d.sigma, d.step_ptr, d.T = c1.data_ptr(), c2.data_ptr(), sg.data_ptr(), step.data_ptr(), T
    d.w, d.seed, d.nan_flag, d.n = w, 0, flag.data_ptr(), n
    d.x_dup0, d.x_dup1, d.t_next, d.t_count, d.done_counter = xin.data_ptr(), xin.data_ptr() + 4 * n, t_next.data_ptr(), 2 * B, done.data_ptr()
    want = x.clone()
    for k, (t_now, t_after) in enumerate(((2, 1), (1, 0), (0, -1))):
        _capi.check(lib.hdiff_ddpm_step_loop(C.byref(d), s), "ddpm_step_loop")
        torch.cuda.synchronize()
        eps = (1. + w) * ec - w * eu
        mean = c1[t_now] * want - c2[t_now] * eps
        want = mean + sg[t_now] * z if t_now > 0 else mean
        assert torch.equal(cur, want), k
        assert torch.equal(xin[:n], want) and torch.equal(xin[n:], want)
        assert int(step.item()) == t_after and t_next.tolist() == [max(t_after, 0)] * (2 * B)
        assert int(done.item()) == 0 and int(flag.item()) == 0          # the counter wrapped back by itself


def test_q_sample_bit_exact_and_clip():
    g = torch.Generator().manual_seed(4)
    B, per = 4, 3 * 16 * 16
    x0, nz = torch.rand(B, per, generator=g) * 2 - 1, torch.randn(B, per, generator=g)
    t = torch.tensor([0, 7, 3, 5])
    sched = O.trainer_schedule(1e-4, 0.028, 8)
    ref = O.q_sample(sched, x0.view(B, 3, 16, 16), t, nz.view(B, 3, 16, 16)).view(B, per)
    sa, sb = sched["sqrt_alphas_bar"].float().to(DEV), sched["sqrt_one_minus_alphas_bar"].float().to(DEV)
    out = torch.empty(B, per, device=DEV)
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    d_x0, d_nz, d_t = x0.to(DEV), nz.to(DEV), t.to(DEV)
    _capi.check(lib.hdiff_q_sample(d_x0.data_ptr(), d_nz.data_ptr(), d_t.data_ptr(), sa.data_ptr(),
                                   sb.data_ptr(), out.data_ptr(), B, per, 8, s))
    assert torch.equal(out.cpu(), ref)
    big = torch.randn(1000, generator=g) * 3
    y = torch.empty(1000, device=DEV)
    d_big = big.to(DEV)
    _capi.check(lib.hdiff_clip(d_big.data_ptr(), y.data_ptr(), C.c_float(-1), C.c_float(1), 1000, s))
    assert torch.equal(y.cpu(), torch.clip(big, -1, 1))


def test_randn_moments_and_determinism():
    n = 1 << 20
    a, b, c = [torch.empty(n, device=DEV) for _ in range(3)]
    lib, s = _capi.lib(), torch.cuda.current_stream().cuda_stream
    _capi.check(lib.hdiff_randn(a.data_ptr(), n, C.c_uint64(42), C.c_uint64(0), s))
    _capi.check(lib.hdiff_randn(b.data_ptr(), n, C.c_uint64(42), C.c_uint64(0), s))
    _capi.check(lib.hdiff_randn(c.data_ptr(), n, C.c_uint64(42), C.c_uint64(1), s))
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1) < 5e-3
    assert abs((a ** 4).mean().item() - 3) < 0.05
    assert abs((a * c).mean().item()) < 5e-3
    assert torch.isfinite(a).all()


@pytest.mark.parametrize("Cc,H,W,B", [(64, 64, 64, 16), (128, 32, 96, 8)])      # >= 192 workgroups per phase launch
def test_upsample_phases_on_the_split_bf16_kernel(Cc, H, W, B, bf16x3_mode):
    """UpSample.forward (ModelCondition.py:85-89): ConvTranspose2d(C, C, 5, 2, 2, 1) as four output-parity phases + Conv3x3.
    At these sizes every phase launch is large enough for the split-bf16 kernel (tap lists + output map, conv3x3_x3.hip):
    it must differ from the fp32-mode result (another program ran) and be fp32-class against float64."""
    lib = bf16x3_mode
    g = torch.Generator().manual_seed(Cc + H)
    x = torch.randn(B, Cc, H, W, generator=g)
    P = {"u.t.weight": torch.randn(Cc, Cc, 5, 5, generator=g) / math.sqrt(Cc * 6.25), "u.t.bias": torch.randn(Cc, generator=g),
         "u.c.weight": torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9), "u.c.bias": torch.randn(Cc, generator=g)}
    Pd = {k: v.to(DEV) for k, v in P.items()}
    xd = x.to(DEV)

    def run():
        plan = E.Plan(DEV)
        y = E.emit_upsample(plan, Pd, "u", xd, B, Cc, H, W)
        plan.pack_weights()
        plan.run()
        torch.cuda.synchronize()
        return y.clone()

    got_x3 = run()
    _capi.check(lib.hdiff_set_contraction_mode(0))
    got_f32 = run()
    _capi.check(lib.hdiff_set_contraction_mode(1))
    assert not torch.equal(got_x3, got_f32), "the split-bf16 kernel did not run"
    u = F.conv_transpose2d(x.double(), P["u.t.weight"].double(), P["u.t.bias"].double(), stride=2, padding=2, output_padding=1)
    want = F.conv2d(u, P["u.c.weight"].double(), P["u.c.bias"].double(), padding=1)
    close(got_x3, want.float(), rel=1e-5, what="upsample split-bf16")
    rms = lambda t: (t.double().cpu() - want).pow(2).mean().sqrt().item()
    assert rms(got_x3) <= 1.5 * rms(got_f32) + 1e-12, (rms(got_x3), rms(got_f32))
