#!/usr/bin/env python3
"""pv_groupnorm_apply (+ SiLU) alone, sustained, by target workgroup count (PV_GN_WGS; one child process per value): the 64 x 64 / 32 x 32 / 16 x 16 level
tensors of a step, single- and two-source (skip concat).  GB/s = (read + write) bytes / time.
usage (GPU box): python tools/diag/gn_apply_ab.py [rounds] [wgs,wgs,...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from photoverse_amd.ops import Recorder, ACT_SILU
dev = torch.device("cuda")
torch.manual_seed(0)
row = []
for (B, hw, c0, c1) in ((16, 4096, 320, 0), (16, 4096, 320, 320), (16, 1024, 640, 0), (16, 1024, 640, 640), (32, 256, 1280, 0), (32, 256, 1280, 1280), (32, 64, 1280, 1280)):
    x = torch.randn(B * hw, c0, device=dev).half()
    x1 = torch.randn(B * hw, c1, device=dev).half() if c1 else None
    C = c0 + c1
    rec = Recorder(dev)
    rec.groupnorm(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), batch=B, hw=hw, x1=x1, act=ACT_SILU)
    rec.run(); torch.cuda.synchronize()
    ap = Recorder.__new__(Recorder); ap.__dict__.update(rec.__dict__); ap.calls = rec.calls[1:]      # the apply launch only
    for _ in range(600): ap.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): ap.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 300 * 1e3
    row.append("%%dx%%dx%%d%%s %%5.1f us %%4.2f TB/s" %% (B, hw, C, "(2 src)" if c1 else "", us, 4.0 * B * hw * C / us / 1e6))
print("  |  ".join(row))
""" % ROOT
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 2
wgs = args[0].split(",") if args else ["4096", "2048", "1024", "512"]
for r in range(rounds):
    for w in wgs:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PV_GN_WGS=w), capture_output=True, text=True)
        print("round %d PV_GN_WGS=%-5s %s" % (r, w, out.stdout.strip() or out.stderr[-300:]), flush=True)
