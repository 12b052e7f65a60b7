"""Micro-benchmark of hdiff_mha_flash_fwd: query tiles per wave (HDIFF_ATT_NQ) A/B'd in one process, interleaved rounds (dev tool)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hdiff_amd  # noqa: E402


def run(variant, shapes, iters=10):
    # variant is read once per process by the library: run each variant in a child process
    code = f"""
import os, sys, ctypes as C
os.environ['HDIFF_ATT_NQ'] = '{variant}'
sys.path.insert(0, {ROOT!r})
import torch, hdiff_amd
lib = hdiff_amd.lib()
s = torch.cuda.current_stream().cuda_stream
for (B, Cc, L) in {shapes!r}:
    qkv = torch.randn(B, 3 * Cc, L, device='cuda')
    o = torch.empty(B, Cc, L, device='cuda')
    for _ in range(2):
        lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), None, B, Cc, 8, L, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range({iters}):
        lib.hdiff_mha_flash_fwd(qkv.data_ptr(), o.data_ptr(), None, B, Cc, 8, L, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / {iters}
    print(f"variant {variant} B={{B}} C={{Cc}} L={{L}}: {{ms:.3f}} ms  {{4.0 * L * L * Cc * B / ms / 1e9:.1f}} TFLOP/s", flush=True)
"""
    subprocess.run([sys.executable, "-c", code], check=True)


if __name__ == "__main__":
    shapes = [(2, 128, 16384), (2, 256, 16384), (1, 128, 65536)]
    variants = sys.argv[1:] or ["4", "8"]
    for rnd in range(2):
        for v in variants:
            run(v, shapes)
