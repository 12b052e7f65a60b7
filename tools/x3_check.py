"""Dev tool: accuracy (vs float64) and speed of the split-bf16 attention forward against the fp32-MFMA kernels."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
lib = hdiff_amd.lib()
s = torch.cuda.current_stream().cuda_stream


def ref64(qkv, heads):
    B, C3, L = qkv.shape
    Cc = C3 // 3
    d = Cc // heads
    q, k, v = [z.reshape(B, heads, d, L).transpose(2, 3).double() for z in qkv.split(Cc, dim=1)]
    w = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(d), dim=-1)
    return (w @ v).transpose(2, 3).reshape(B, Cc, L)


import ctypes as C


def workspace(B, Cc, heads, L):
    need = C.c_int64(0)
    assert lib.hdiff_mha_flash_fwd_workspace(B, Cc, heads, L, C.byref(need)) == 0
    return (torch.empty(need.value // 4 + 1, device="cuda"), need.value) if need.value > 0 else (None, 0)


def fwd(qkv, o, B, Cc, heads, L, ws):
    """ws = (tensor, bytes): the pre-split kernel in bf16x3 mode; (None, 0): the kernels that split in their loop"""
    return lib.hdiff_mha_flash_fwd_ws(qkv.data_ptr(), o.data_ptr(), None, B, Cc, heads, L,
                                      None if ws[0] is None else ws[0].data_ptr(), ws[1], s)


def run(mode, qkv, heads, presplit=True):
    B, C3, L = qkv.shape
    lib.hdiff_set_contraction_mode(mode)
    o = torch.empty(B, C3 // 3, L, device="cuda")
    rc = fwd(qkv, o, B, C3 // 3, heads, L, workspace(B, C3 // 3, heads, L) if presplit else (None, 0))
    assert rc == 0, lib.hdiff_last_error()
    torch.cuda.synchronize()
    return o


for d, L, scale in [(16, 2048, 1.0), (16, 4096, 3.0), (16, 4096, -1.0), (16, 8192, -2.0), (32, 2048, 1.0), (32, 1024, 2.0)]:
    g = torch.Generator().manual_seed(d + L)
    qkv = torch.randn(1, 3 * 8 * d, L, generator=g)
    if scale == -1.0:       # the row maximum keeps moving: key magnitudes ramp up along the sequence
        qkv[:, 8 * d:16 * d] *= torch.linspace(0.3, 5.0, L)
    elif scale == -2.0:     # wide-range V channels, a few huge keys late in the sequence
        qkv[:, 16 * d:] *= torch.exp(torch.randn(1, 8 * d, 1, generator=g) * 6) * torch.exp(torch.randn(1, 8 * d, L, generator=g) * 2)
        qkv[:, 8 * d:16 * d, 5000:5003] *= 12.0
    else:
        qkv = qkv * scale
    qkv = qkv.cuda()
    r = ref64(qkv, 8)
    for mode, name, pre in [(0, "f32        ", False), (1, "bf16x3 loop", False), (1, "bf16x3 pre ", True)]:
        o = run(mode, qkv, 8, pre).double()
        err = (o - r).abs()
        print(f"d={d} L={L} scale={scale} {name}: max abs err {err.max().item():.3e}  rms err {err.pow(2).mean().sqrt().item():.3e}"
              f"  (rms of ref {r.pow(2).mean().sqrt().item():.3e})", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "time":
    for (B, Cc, L) in [(int(sys.argv[2]) if len(sys.argv) > 2 else 1, 128, 65536), (2, 256, 16384), (4, 256, 4096)]:
        qkv = torch.randn(B, 3 * Cc, L, device="cuda")
        o = torch.empty(B, Cc, L, device="cuda")
        for mode, name, pre in [(0, "f32        ", False), (1, "bf16x3 loop", False), (1, "bf16x3 pre ", True)]:
            lib.hdiff_set_contraction_mode(mode)
            ws = workspace(B, Cc, 8, L) if pre else (None, 0)
            for _ in range(2):
                fwd(qkv, o, B, Cc, 8, L, ws)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fwd(qkv, o, B, Cc, 8, L, ws)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"{name} B={B} C={Cc} L={L}: {ms:.3f} ms  {4.0 * L * L * Cc * B / ms / 1e9:.1f} TFLOP/s (fp32-equivalent)", flush=True)
