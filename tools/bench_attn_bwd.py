"""Dev tool: time hdiff_mha_flash_bwd (delta + fused five-product kernel + dQ slab reduce) at the training shapes.
  python tools/bench_attn_bwd.py [B]      prints ms and algorithmic TFLOP/s (10 * L^2 * C * B FLOP per launch)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hdiff_amd  # noqa: E402

lib = hdiff_amd.lib()
s = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for (Cc, L) in [(128, 65536), (256, 16384), (256, 4096), (256, 1024)]:
    qkv = torch.randn(B, 3 * Cc, L, device="cuda")
    d_o = torch.randn(B, Cc, L, device="cuda")
    o = torch.empty(B, Cc, L, device="cuda")
    lse = torch.empty(B, 8, L, device="cuda")
    delta = torch.empty(B, 8, L, device="cuda")
    dqkv = torch.empty_like(qkv)
    need = C.c_int64(0)
    lib.hdiff_mha_flash_bwd_workspace(B, Cc, 8, L, C.byref(need))
    ws = torch.empty(max(need.value, 1), device="cuda")
    lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Cc, 8, L, s)

    def run():
        rc = lib.hdiff_mha_flash_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                     dqkv.data_ptr(), ws.data_ptr(), B, Cc, 8, L, s)
        assert rc == 0, lib.hdiff_last_error()
    run()
    torch.cuda.synchronize()
    iters = 3 if L >= 65536 else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"bwd B={B} C={Cc} L={L}: {ms:.3f} ms  {10.0 * L * L * Cc * B / ms / 1e9:.1f} TFLOP/s (5 products)  ws {need.value * 4 / 1e9:.2f} GB",
          flush=True)
