"""Dev tool: launch the dominant attention kernel a few times at the bench shape (for rocprofv3 --pmc passes).
argv: [batch] [channels = 128] [L = 65536]; the contraction mode comes from HDIFF_CONTRACT (default bf16x3: the pre-split
kernels, with their workspace)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
lib = hdiff_amd.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
qkv = torch.randn(B, 3 * Cc, L, device="cuda")
o = torch.empty(B, Cc, L, device="cuda")
need = C.c_int64(0)
lib.hdiff_mha_flash_fwd_workspace(B, Cc, 8, L, C.byref(need))
ws = torch.empty(need.value // 4 + 1, device="cuda") if need.value > 0 else None
s = torch.cuda.current_stream().cuda_stream
def go():
    lib.hdiff_mha_flash_fwd_ws(qkv.data_ptr(), o.data_ptr(), None, B, Cc, 8, L, None if ws is None else ws.data_ptr(), need.value, s)
go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    go()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"fwd B={B} C={Cc} L={L}: {ms:.3f} ms  {4.0 * L * L * Cc * B / ms / 1e9:.1f} TFLOP/s-eq (split pass included)")
