"""Dev tool (round 5): the error-class margin of the split-operand attention forwards -- rms and worst error against float64 as a
ratio to the fp32-MFMA kernel's (the gate of tests/test_gpu_ops.py::test_flash_attention_split_bf16_is_fp32_class is 1.25x / 2x).
   python3 tools/attn_error_ratio.py            (HDIFF_LIB selects a variant library)"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
lib = hdiff_amd.lib(); s = torch.cuda.current_stream().cuda_stream
def ref64(qkv, heads):
    B, C3, L = qkv.shape; Cc = C3 // 3; d = Cc // heads
    q, k, v = [z.reshape(B, heads, d, L).transpose(2, 3).double() for z in qkv.split(Cc, dim=1)]
    w = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(d), dim=-1)
    return (w @ v).transpose(2, 3).reshape(B, Cc, L)
def run(qkv, heads, mode):
    lib.hdiff_set_contraction_mode(mode)
    B, C3, L = qkv.shape; Cc = C3 // 3
    need = C.c_int64(0); lib.hdiff_mha_flash_fwd_workspace(B, Cc, heads, L, C.byref(need))
    ws = torch.empty(need.value // 4 + 1, device="cuda"); o = torch.empty(B, Cc, L, device="cuda")
    assert lib.hdiff_mha_flash_fwd_ws(qkv.data_ptr(), o.data_ptr(), None, B, Cc, heads, L, ws.data_ptr(), need.value, s) == 0
    torch.cuda.synchronize(); return o
for d, L, B, scale in [(16, 1024, 2, 1.0), (16, 4096, 1, 3.0), (16, 8192, 1, 0.3), (32, 2048, 1, 1.0), (32, 512, 2, 2.0), (32, 4096, 1, 3.0), (32, 8192, 1, 0.3), (32, 16384, 1, 1.0)]:
    g = torch.Generator().manual_seed(100 + d + L)
    qkv = (torch.randn(B, 3 * 8 * d, L, generator=g) * scale).cuda()
    r = ref64(qkv, 8)
    e = {m: (run(qkv, 8, m).double() - r) for m in (0, 1)}
    rms = {m: e[m].pow(2).mean().sqrt().item() for m in e}; worst = {m: e[m].abs().max().item() for m in e}
    print(f"d {d:2d} L {L:5d} B {B} scale {scale}: rms {rms[1]:.3e} / fp32 kernel {rms[0]:.3e} = {rms[1] / rms[0]:.3f}   worst {worst[1]:.3e} / {worst[0]:.3e} = {worst[1] / worst[0]:.3f}")
lib.hdiff_set_contraction_mode(1)
