"""Dev tool (round 6): run a command and sample the engine clock / board power of this process's GPU from sysfs while it runs
  python tools/power_watch.py <command ...>     (bench.ClockSampler; the child is an ordinary subprocess)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import ClockSampler  # noqa: E402
c = ClockSampler(0)
with c:
    r = subprocess.run(sys.argv[1:], capture_output=True, text=True, timeout=300)
s = c.summary()
print(r.stdout.strip(), f"| sclk {s.get('sclk_mhz_mean')} MHz (min {s.get('sclk_mhz_min')}), board {s.get('board_power_w_mean')} W", flush=True)
if r.returncode != 0:
    print(r.stderr[-2000:])
