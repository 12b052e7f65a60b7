"""Dev tool (round 5): in-kernel clock and cycles per stage of the d_head 16 forward (attention_h2w.hip built with -DH2W_DIAG=1:
tools/scripts/ab_build.sh h2w_diag attention_h2w -DH2W_DIAG=1; run with HDIFF_LIB=tools/bin/libhdiff_h2w_diag.so).
Every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its tile loop into a spare part of the
workspace; this prints the median clock and the cycles per (wave, stage) -- wall time alone cannot tell a slower body from a
lower clock (MI355X_MICROARCH.md, DVFS give-back).  argv: [batch = 16] [L = 65536] [seconds of warm launches = 2]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, hdiff_amd
lib = hdiff_amd.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
warm = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
Cc, heads = 128, 8
qkv = torch.randn(B, 3 * Cc, L, device="cuda")
o = torch.empty(B, Cc, L, device="cuda")
need = C.c_int64(0)
lib.hdiff_mha_flash_fwd_workspace(B, Cc, heads, L, C.byref(need))
ws = torch.zeros(need.value // 8 + 1, device="cuda", dtype=torch.int64)
s = torch.cuda.current_stream().cuda_stream
go = lambda: lib.hdiff_mha_flash_fwd_ws(qkv.data_ptr(), o.data_ptr(), None, B, Cc, heads, L, ws.data_ptr(), need.value, s)
go(); torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < warm:
    go(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); go(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
pair_words = 288 * L // 8
w = ws[: B * heads * pair_words].view(B * heads, pair_words)
off = (256 * L + 64) // 8
d = w[:, off: off + 2 * (L // 256)].reshape(-1, 2).cpu().double()
d = d[(d[:, 0] > 0) & (d[:, 1] > 0)]
if len(d) == 0:
    print(f"fwd B={B} L={L}: {ms:.3f} ms (no stamps: not a -DH2W_DIAG build)")
else:
    clk = (d[:, 0] / d[:, 1] * 100.0).median().item()
    stages = 2 * (L // 32)
    cyc = (d[:, 0] / stages).median().item()
    print(f"fwd B={B} L={L}: {ms:.3f} ms; in-kernel clock {clk:.0f} MHz (median of {len(d)} workgroups), {cyc:.1f} cycles per stage and wave "
          f"= {cyc / clk * 1e3:.1f} ns; workgroup loop {d[:, 0].median().item() / clk / 1e3:.3f} ms")
