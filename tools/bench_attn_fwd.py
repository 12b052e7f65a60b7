import sys, torch
sys.path.insert(0, '/root/repo')
import hdiff_amd
lib = hdiff_amd.lib(); s = torch.cuda.current_stream().cuda_stream
for (B, Cc, L) in [(16, 128, 65536), (16, 256, 16384)]:
    qkv = torch.randn(B, 3 * Cc, L, device='cuda'); o = torch.empty(B, Cc, L, device='cuda')
    for _ in range(2): lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), None, B, Cc, 8, L, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 4 if L > 20000 else 10
    for _ in range(n): lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), None, B, Cc, 8, L, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"fwd B={B} C={Cc} L={L}: {ms:.3f} ms  {4.0 * L * L * Cc * B / ms / 1e9:.1f} TFLOP/s", flush=True)
