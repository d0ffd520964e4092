#!/usr/bin/env python3
"""In-kernel stamps of attn8_kernel (pv_attn.hip built with -DPV_ATTN8_STAMPS into a private library under /tmp): per wave, the shader cycles
of the four parts of a tile - vector segment (softmax), wait at the barrier behind it, matrix segment (P.V + K.Q^T + fragment reads + DMA issue),
wait at the barrier behind that - averaged over the steady-state tiles of three workgroups, and the in-kernel clock.

usage (GPU box): python tools/diag/attn8_stamps.py [variants, comma separated]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r"""
import ctypes, os, sys, torch
sys.path.insert(0, %r)
from photoverse_amd import _lib
_lib.LIB = "/tmp/libpv_attn8_stamps.so"
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, n, d = 16, 4096, 40
C = 8 * d
torch.manual_seed(0)
qkv = torch.randn(B * n, 3 * C, device=dev).half()
rec = Recorder(dev)
rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=8, nq=n, nk=n, d=d)
for _ in range(200):          # ~0.15 s of back-to-back launches: the clock settles
    rec.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    rec.run()
e1.record()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (3 * 8 * 8))()
fn = rec.lib.pv_attn8_read_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
print("variant %%s: launch %%.1f us (stamped build)" %% (os.environ["PV_ATTN8"], e0.elapsed_time(e1) / 20 * 1e3))
print("  wg wave |  vector  wait-B  matrix  wait-A | per tile |  clock GHz")
for wg in range(3):
    for w in range(8):
        v = [out[(wg * 8 + w) * 8 + k] for k in range(8)]
        nt = max(v[6], 1)
        per = [x / nt for x in v[:4]]
        clk = v[4] / max(v[5], 1) * 0.1
        print("  %%2d  %%d   | %%7.0f %%7.0f %%7.0f %%7.0f | %%7.0f  |  %%.2f" %% (wg, w, per[0], per[1], per[2], per[3], sum(per), clk))
""" % ROOT


def main():
    import photoverse_amd.build as b
    variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["1"]
    objs = []
    for f in b.SOURCES:
        o = "/tmp/a8st_%s.o" % f
        extra = ["-DPV_ATTN8_STAMPS"] if f == "pv_attn.hip" else []
        subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), *extra, "-c", os.path.join(b.CSRC, f), "-o", o])
        objs.append(o)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libpv_attn8_stamps.so", *objs])
    for v in variants:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PV_ATTN8=v), capture_output=True, text=True, timeout=600)
        print(r.stdout + (r.stderr[-3000:] if r.returncode else ""), flush=True)


if __name__ == "__main__":
    main()
