"""Shared by tests/test_gpu_backward.py and tools/attn_bwd_error_ratio.py: inputs that stress the fp16-pair attention backward
(attention_bwd_h2.hip scales V by one power of two per (sample, head), dO by one per head and one per query position), a query-chunked float64 reference of one
(sample, head) pair, and the error statistics of the split-operand kernel beside the fp32-input kernel's."""
import ctypes as C
import math

import torch

DEV = "cuda:0"
CASES = ("plain", "loud-dO-pixel", "wide-V", "peaked", "tiny-dO", "zero-dO-head", "dO-1e-30")


def make_case(name, d, L, B, heads, g):
    Cc = heads * d
    sc = 3.0 if name == "peaked" else 1.3
    qkv = torch.randn(B, 3 * Cc, L, generator=g) * sc
    d_o = torch.randn(B, Cc, L, generator=g)
    if name == "loud-dO-pixel": d_o[:, :, L // 3] *= 1e4                  # one position 1e4 x the rest: everything else sits 13 binades below the scale
    if name == "wide-V": qkv[:, 2 * Cc:] *= (2.0 ** torch.linspace(-15, 15, Cc))[None, :, None]     # V channels 2^30 apart
    if name == "tiny-dO": d_o *= 1e-20
    if name == "dO-1e-30": d_o *= 1e-30                                  # the head's scale 2^so reaches 2^111: 2^(so + t_q) of a quiet row would leave fp32
    if name == "zero-dO-head":                                            # a masked / zero-weight sample: one (sample, head) of dO all zero (so = 113),
        d_o[0, 2 * d:3 * d] = 0.0                                         # and a run of silent positions inside a live head (t_q at its clamp of 24)
        d_o[B - 1, :d, 100:200] = 0.0
    return qkv, d_o


def run_bwd(lib, qkv, d_o, heads, mode):
    """forward + backward through the C ABI in contraction mode `mode` (0 = fp32-input MFMA, 1 = split operands); returns dqkv"""
    s = torch.cuda.current_stream().cuda_stream
    before = lib.hdiff_get_contraction_mode()
    try:
        assert lib.hdiff_set_contraction_mode(mode) == 0
        B, C3, L = qkv.shape; Cc = C3 // 3
        o = torch.empty(B, Cc, L, device=DEV); lse = torch.empty(B, heads, L, device=DEV)
        assert lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Cc, heads, L, s) == 0
        delta = torch.empty(B, heads, L, device=DEV); dqkv = torch.full_like(qkv, float("nan"))
        need = C.c_int64(0); assert lib.hdiff_mha_flash_bwd_workspace(B, Cc, heads, L, C.byref(need)) == 0
        ws = torch.full((max(need.value, 1),), float("nan"), device=DEV)
        assert lib.hdiff_mha_flash_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(),
                                       ws.data_ptr(), B, Cc, heads, L, s) == 0
        torch.cuda.synchronize()
    finally:
        lib.hdiff_set_contraction_mode(before)
    return dqkv


def ref64(qkv, d_o, b, h, d, Cc, chunk=4096):
    """float64 gradients of one (sample, head), query-chunked so that L = 65536 fits"""
    sl = slice(h * d, (h + 1) * d)
    Q = qkv[b, sl].double().t(); K = qkv[b, Cc + h * d:Cc + (h + 1) * d].double().t()
    V = qkv[b, 2 * Cc + h * d:2 * Cc + (h + 1) * d].double().t(); dO = d_o[b, sl].double().t()
    L = Q.shape[0]; dQ = torch.empty_like(Q); dK = torch.zeros_like(K); dV = torch.zeros_like(V)
    for i in range(0, L, chunk):
        P = torch.softmax(Q[i:i + chunk] @ K.t() / math.sqrt(d), dim=-1)
        dP = dO[i:i + chunk] @ V.t()
        dS = P * (dP - (dO[i:i + chunk] * (P @ V)).sum(-1, keepdim=True))
        dQ[i:i + chunk] = dS @ K / math.sqrt(d); dK += dS.t() @ Q[i:i + chunk] / math.sqrt(d); dV += P.t() @ dO[i:i + chunk]
    return {"dQ": dQ.t(), "dK": dK.t(), "dV": dV.t()}


def error_stats(lib, qkv, d_o, heads, pairs=None):
    """{name: {"pair": [(b, h, rms_h2, rms_f32, worst_h2, worst_f32, mag)], "rms": (h2, f32) over all pairs, "worst": (h2, f32), "mag"}}"""
    B, C3, L = qkv.shape; Cc = C3 // 3; d = Cc // heads
    g32, gh2 = run_bwd(lib, qkv, d_o, heads, 0), run_bwd(lib, qkv, d_o, heads, 1)
    assert torch.isfinite(gh2).all() and not torch.equal(gh2, g32), "the split-operand backward did not run"
    out = {n: {"pair": [], "sq": [0.0, 0.0], "worst": [0.0, 0.0], "mag": 0.0, "n": 0} for n in ("dQ", "dK", "dV")}
    for b, h in (pairs if pairs is not None else [(b, h) for b in range(B) for h in range(heads)]):
        ref = ref64(qkv, d_o, b, h, d, Cc)
        for i, n in enumerate(("dQ", "dK", "dV")):
            rows = slice(i * Cc + h * d, i * Cc + (h + 1) * d)
            e2, e0 = gh2[b, rows].double() - ref[n], g32[b, rows].double() - ref[n]
            st = out[n]
            st["pair"].append((b, h, e2.pow(2).mean().sqrt().item(), e0.pow(2).mean().sqrt().item(), e2.abs().max().item(), e0.abs().max().item(),
                               ref[n].abs().max().item()))
            st["sq"][0] += e2.pow(2).sum().item(); st["sq"][1] += e0.pow(2).sum().item(); st["n"] += e2.numel()
            st["worst"] = [max(st["worst"][0], e2.abs().max().item()), max(st["worst"][1], e0.abs().max().item())]
            st["mag"] = max(st["mag"], ref[n].abs().max().item())
    for st in out.values():
        st["rms"] = tuple(math.sqrt(v / st["n"]) for v in st["sq"])
    return out
