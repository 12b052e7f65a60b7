"""Dev tool: launch the attention backward a few times at the training shape (for rocprofv3 --pmc passes).  argv: [B = 4] [channels = 128] [L = 65536]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
lib = hdiff_amd.lib(); s = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
qkv = torch.randn(B, 3 * Cc, L, device="cuda"); d_o = torch.randn(B, Cc, L, device="cuda")
o = torch.empty(B, Cc, L, device="cuda"); lse = torch.empty(B, 8, L, device="cuda"); delta = torch.empty(B, 8, L, device="cuda")
dqkv = torch.empty_like(qkv); need = C.c_int64(0)
lib.hdiff_mha_flash_bwd_workspace(B, Cc, 8, L, C.byref(need)); ws = torch.empty(max(need.value, 1), device="cuda")
lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Cc, 8, L, s)
for _ in range(2):
    assert lib.hdiff_mha_flash_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(),
                                   ws.data_ptr(), B, Cc, 8, L, s) == 0
torch.cuda.synchronize(); print("done")
