#!/usr/bin/env python3
"""In-kernel stamps of attn8_bwd_kernel (pv_attnbwd.hip built with -DPV_ATTN8_BWD_STAMPS into a private library under /tmp): per wave and pass,
the shader cycles of the four parts of a 32-row step - vector segment (exponentials, dS), wait at the barrier behind it, matrix segment (gradient
products + next step's S / dP + DMA issue), wait behind that - averaged over the steady-state steps of workgroup 300, and the in-kernel clock.

usage (GPU box): python tools/diag/attn8_bwd_stamps.py [variants, comma separated]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r"""
import ctypes, os, sys, torch
sys.path.insert(0, %r)
from photoverse_amd import _lib
_lib.LIB = "/tmp/libpv_attn8_bwd_stamps.so"
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, n, d, H = 16, 4096, 40, 8
C = H * d
torch.manual_seed(0)
qkv = torch.randn(B * n, 3 * C, device=dev).half()
do = torch.randn(B * n, C, device=dev).half()
pre = Recorder(dev)
lse = torch.empty((B, H, n), dtype=torch.float32, device=dev)
o = pre.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=H, nq=n, nk=n, d=d, lse=lse)
pre.run()
rec = Recorder(dev)
rec.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, batch=B, heads=H, nq=n, nk=n, d=d)
for _ in range(100):
    rec.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    rec.run()
e1.record()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (2 * 8 * 8))()
fn = rec.lib.pv_attn8_bwd_read_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
print("variant %%s: prep + dKV + dQ %%.1f us (stamped build)" %% (os.environ["PV_ATTN8_BWD"], e0.elapsed_time(e1) / 20 * 1e3))
print("  pass wave |  vector  wait-B  matrix  wait-A | per step |  clock GHz")
for ps in range(2):
    for w in range(8):
        v = [out[(ps * 8 + w) * 8 + k] for k in range(8)]
        nt = max(v[6], 1)
        per = [x / nt for x in v[:4]]
        clk = v[4] / max(v[5], 1) * 0.1
        print("  %%s  %%d   | %%7.0f %%7.0f %%7.0f %%7.0f | %%7.0f  |  %%.2f" %% ("dKV" if ps else "dQ ", w, per[0], per[1], per[2], per[3], sum(per), clk))
""" % ROOT


def main():
    import photoverse_amd.build as b
    variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "81"]
    objs = []
    for f in b.SOURCES:
        o = "/tmp/b8st_%s.o" % f
        extra = ["-DPV_ATTN8_BWD_STAMPS"] if f == "pv_attnbwd.hip" else []
        subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), *extra, "-c", os.path.join(b.CSRC, f), "-o", o])
        objs.append(o)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libpv_attn8_bwd_stamps.so", *objs])
    for v in variants:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PV_ATTN8_BWD=v), capture_output=True, text=True, timeout=600)
        print(r.stdout + (r.stderr[-3000:] if r.returncode else ""), flush=True)


if __name__ == "__main__":
    main()
