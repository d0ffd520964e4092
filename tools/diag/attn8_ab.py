#!/usr/bin/env python3
"""Self-attention d = 40: the 4-wave kernel against the variants of the 8-wave staggered kernel (PV_ATTN8), one process per variant
(the switch is read once per process), several rounds on ONE box.  Each child checks the variant against fp32 SDPA and prints a
checksum of the fp16 output, so that bit-identity between variants is visible in the log.

usage (GPU box): python tools/diag/attn8_ab.py [rounds] [variants, comma separated; -1 = the 4-wave kernel]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r"""
import os, sys, hashlib, torch
import torch.nn.functional as F
sys.path.insert(0, %r)
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
torch.manual_seed(0)

def run(B, n, spiky=False, reps=0):
    H, d = 8, 40
    C = H * d
    g = torch.Generator().manual_seed(n + B)
    qkv = torch.randn(B * n, 3 * C, generator=g).half()
    if spiky:
        qkv[(n * 3) // 5, C:2 * C] *= 12.0
    x = qkv.cuda()
    rec = Recorder(dev)
    out = rec.attention(x[:, :C], x[:, C:2 * C], x[:, 2 * C:], batch=B, heads=H, nq=n, nk=n, d=d)
    rec.run()
    torch.cuda.synchronize()
    if reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(600):          # ~0.3 s of back-to-back launches first: a 12-ms sample from idle reads 10 percent slow (clock ramp)
            rec.run()
        e0.record()
        for _ in range(reps):
            rec.run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    q, k, v = [t.float().view(B, n, H, d).transpose(1, 2) for t in qkv.split(C, dim=1)]
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * n, C)
    o = out.float().cpu()
    err = ((o - ref).norm() / ref.norm()).item()
    return err, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]

mode = sys.argv[1]
if mode == "check":
    for (B, n, sp) in ((4, 4096, False), (4, 4096, True), (4, 4000, False), (8, 2304, False)):
        err, h = run(B, n, sp)
        print("check B=%%d n=%%d spiky=%%d  rel-L2 %%.3e  sha %%s" %% (B, n, sp, err, h), flush=True)
        assert err < 2e-3
else:
    us = run(16, 4096, reps=200)
    print("time %%.1f us  (%%.0f TFLOP/s, %%.3f of 2.5 PF)" %% (us, 4.0 * 16 * 4096 * 4096 * 320 / us / 1e6, 4.0 * 16 * 4096 * 4096 * 320 / us / 1e6 / 2500), flush=True)
""" % ROOT


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["-1", "1", "9", "225", "497"])]
    for v in variants:
        env = dict(os.environ, PV_ATTN8=str(v))
        r = subprocess.run([sys.executable, "-c", CHILD, "check"], env=env, capture_output=True, text=True, timeout=600)
        print("== variant %d  (check)\n%s%s" % (v, r.stdout, r.stderr[-2000:] if r.returncode else ""), flush=True)
    for i in range(rounds):
        for v in variants:
            env = dict(os.environ, PV_ATTN8=str(v))
            r = subprocess.run([sys.executable, "-c", CHILD, "time"], env=env, capture_output=True, text=True, timeout=600)
            print("round %d  variant %2d  %s" % (i, v, r.stdout.strip() or r.stderr[-500:]), flush=True)


if __name__ == "__main__":
    main()
