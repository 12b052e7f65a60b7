"""Dev tool (round 6): what clock and board power does ONE hot kernel hold when it runs back to back?
  python tools/kernel_power.py bwd16|bwd32|fwd16|fwd32|conv3 [batch] [seconds]      (conv3: the 128 -> 128 3x3 convolution at 256x256 behind GroupNorm, fp16 pairs)
Runs the launch in a loop for `seconds` (default 4) after a 1 s warm-up, samples the engine clock and the board power from sysfs at
20 Hz (bench.ClockSampler), prints ms per launch, MHz and W.  A kernel that sits on the board's power limit trades cycles for clock:
an instruction-count saving then shows as a higher clock at the same wall time -- read this BEFORE believing a cycle model."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hdiff_amd  # noqa: E402
from bench import ClockSampler  # noqa: E402

lib = hdiff_amd.lib()
s = torch.cuda.current_stream().cuda_stream
what = sys.argv[1] if len(sys.argv) > 1 else "bwd16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if what.startswith("fwd") else 4)
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
if what == "conv3":
    import math
    from hdiff_amd import engine as E
    Cin = Cout = 128; S = 256
    x = torch.randn(B, Cin, S, S, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, device="cuda")
    plan = E.Plan("cuda:0")
    pk = E._std_pack(plan, w, 3, 1)
    out = plan.buf(B, Cout, S, S)
    gamma, beta = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.3
    pre = E.Plan("cuda:0")
    g = pre.gn_scale_shift(x, None, gamma, beta, B, S * S)
    pre.run(); torch.cuda.synchronize()
    plan._gn_src[id(g[0])] = (gamma, beta, (Cin // 32) * S * S, 1.0)
    plan.conv(x, None, pk, bias, out, B=B, H=S, W=S, VH=S, VW=S, gn=g)
    plan.pack_weights()
    run_conv = plan.run
Cc, L = (128, 65536) if what.endswith("16") else (256, 16384)
if what != "conv3":
    qkv = torch.randn(B, 3 * Cc, L, device="cuda")
    o = torch.empty(B, Cc, L, device="cuda")
    lse = torch.empty(B, 8, L, device="cuda")
if what == "conv3":
    def run():
        for _ in range(50):
            run_conv()
elif what.startswith("fwd"):
    need = C.c_int64(0)
    lib.hdiff_mha_flash_fwd_workspace(B, Cc, 8, L, C.byref(need))
    ws = torch.empty(max(need.value, 16), dtype=torch.uint8, device="cuda")

    def run():
        rc = lib.hdiff_mha_flash_fwd_ws(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Cc, 8, L, ws.data_ptr(), need.value, s)
        assert rc == 0, lib.hdiff_last_error()
else:
    d_o = torch.randn(B, Cc, L, device="cuda")
    delta = torch.empty(B, 8, L, device="cuda")
    dqkv = torch.empty_like(qkv)
    need = C.c_int64(0)
    lib.hdiff_mha_flash_bwd_workspace(B, Cc, 8, L, C.byref(need))
    ws = torch.empty(max(need.value, 1), device="cuda")
    lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, Cc, 8, L, s)

    def run():
        rc = lib.hdiff_mha_flash_bwd(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(),
                                     ws.data_ptr(), B, Cc, 8, L, s)
        assert rc == 0, lib.hdiff_last_error()

t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    run()
    torch.cuda.synchronize()
clock = ClockSampler(0)
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with clock:
    t0 = time.perf_counter()
    e0.record()
    while time.perf_counter() - t0 < secs:
        run()
        run()
        torch.cuda.synchronize()
        n += 2
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
c = clock.summary()
if what == "conv3":
    ms /= 50
print(f"{what} B={B} C={Cc} L={L}: {ms:.3f} ms per launch over {n} launches; sclk {c.get('sclk_mhz_mean')} MHz "
      f"(min {c.get('sclk_mhz_min')}, max {c.get('sclk_mhz_max')}), board {c.get('board_power_w_mean')} W", flush=True)
