"""Dev tool: launch the dominant attention kernel a few times at the bench shape (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
lib = hdiff_amd.lib()
B, Cc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 128, 65536
qkv = torch.randn(B, 3 * Cc, L, device="cuda")
o = torch.empty(B, Cc, L, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), None, B, Cc, 8, L, s)
torch.cuda.synchronize()
print("done")
