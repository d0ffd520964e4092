#!/usr/bin/env python3
"""Board power / clocks while one kernel shape loops (rocm-smi sampled from a side thread).
Usage (GPU box): python3 tools/power_probe.py {conv|attn|gn|idle} [seconds]"""
import os, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photoverse_amd.ops import Recorder

dev = torch.device("cuda")
what = sys.argv[1] if len(sys.argv) > 1 else "conv"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
B = 16
rec = Recorder(dev)
if what.startswith("conv"):
    zero = what == "conv0"
    x = (torch.zeros if zero else torch.randn)(B * 64 * 64, 320, device=dev).half()
    w = ((torch.zeros if zero else torch.randn)(320, 2880, device=dev) * 0.02).half()
    for _ in range(20):
        rec.gemm(x, w, bias=torch.zeros(320, device=dev), conv=dict(batch=B, hin=64, win=64, hout=64, wout=64))
elif what == "attn":
    qkv = torch.randn(B * 4096, 960, device=dev).half()
    for _ in range(4):
        rec.attention(qkv[:, :320], qkv[:, 320:640], qkv[:, 640:], batch=B, heads=8, nq=4096, nk=4096, d=40)
elif what == "gn":
    x = torch.randn(B * 4096, 320, device=dev).half()
    for _ in range(40):
        rec.groupnorm(x, torch.ones(320, device=dev), torch.zeros(320, device=dev), batch=B, hw=4096, act=1)

samples, stop = [], False
def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True)
        keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "fclk", "junction", "Temperature (Sensor junction)"))]
        samples.append((time.time(), keep))
        time.sleep(0.5)
t = threading.Thread(target=sampler); t.start()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    if what != "idle":
        rec.run(); n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    else:
        time.sleep(0.1)
torch.cuda.synchronize()
dt = time.time() - t0
stop = True; t.join()
print(f"== {what}: {n} replays in {dt:.1f} s" + (f" -> {dt / max(n, 1) / len(rec) * 1e6:.1f} us per launch" if n else ""))
for ts, keep in samples[2:8]:
    print(f"  t+{ts - t0:4.1f}s  " + " | ".join(keep))
