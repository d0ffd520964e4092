#!/usr/bin/env python3
"""Self-attention backward d = 40: the 4-wave passes of pv_train.hip (PV_ATTN8_BWD=-1) against the 8-wave staggered passes of pv_attnbwd.hip
(variant bits: 1 stagger, 16 s_setprio 1 in the matrix segments, 64 fenced vector-segment passes; instantiated: 0, 1, 65, 81), one process per variant, several rounds on ONE box, sustained timing.

usage (GPU box): python tools/diag/attn8_bwd_ab.py [rounds] [variants, comma separated; -1 = the 4-wave kernels] [batch] [n] [d]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, %r)
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, n, H, d = int(sys.argv[2]), int(sys.argv[3]), 8, int(sys.argv[4])
C = H * d
g = torch.Generator().manual_seed(7)
qkv = torch.randn(B * n, 3 * C, generator=g).half().cuda()
do = torch.randn(B * n, C, generator=g).half().cuda()
pre = Recorder(dev)
lse = torch.empty((B, H, n), dtype=torch.float32, device=dev)
o = pre.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=H, nq=n, nk=n, d=d, lse=lse)
pre.run()
rec = Recorder(dev)
dq, dk, dv = rec.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, batch=B, heads=H, nq=n, nk=n, d=d)
rec.run()
torch.cuda.synchronize()
if sys.argv[1] == "check":
    b = min(B, 2)
    q32, k32, v32 = (qkv[:b * n, i * C:(i + 1) * C].float().view(b, n, H, d).transpose(1, 2).clone().requires_grad_() for i in range(3))
    ref = F.scaled_dot_product_attention(q32, k32, v32)
    ref.backward(do[:b * n].float().view(b, n, H, d).transpose(1, 2))
    for name, got, t in zip("qkv", (dq, dk, dv), (q32, k32, v32)):
        want = t.grad.transpose(1, 2).reshape(b * n, C)
        print("check d%%s rel-L2 %%.3e" %% (name, ((got[:b * n].float() - want).norm() / want.norm()).item()), flush=True)
else:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(150):
        rec.run()
    e0.record()
    for _ in range(100):
        rec.run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 100 * 1e3
    fl = 10.0 * B * H * n * n * d
    print("time %%.1f us  (%%.0f TFLOP/s counted, %%.3f of 2.5 PF)" %% (us, fl / us / 1e6, fl / us / 1e6 / 2500), flush=True)
""" % ROOT


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["-1", "0", "1", "65", "81"])]
    B = sys.argv[3] if len(sys.argv) > 3 else "16"
    n = sys.argv[4] if len(sys.argv) > 4 else "4096"
    d = sys.argv[5] if len(sys.argv) > 5 else "40"
    for v in variants:
        env = dict(os.environ, PV_ATTN8_BWD=str(v))
        r = subprocess.run([sys.executable, "-c", CHILD, "check", B, n, d], env=env, capture_output=True, text=True, timeout=900)
        print("== variant %d  (check)\n%s%s" % (v, r.stdout, r.stderr[-2000:] if r.returncode else ""), flush=True)
    for i in range(rounds):
        for v in variants:
            env = dict(os.environ, PV_ATTN8_BWD=str(v))
            r = subprocess.run([sys.executable, "-c", CHILD, "time", B, n, d], env=env, capture_output=True, text=True, timeout=900)
            print("round %d  variant %2d  %s" % (i, v, r.stdout.strip() or r.stderr[-500:]), flush=True)


if __name__ == "__main__":
    main()
