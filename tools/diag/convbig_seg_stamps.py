#!/usr/bin/env python3
"""In-kernel stamps of pv_convbig.hip's 256-row tile (built with -DPV_CONVBIG_STAMPS into a private library under /tmp): per wave, the shader
cycles per stage of the LOAD segment, the wait at the barrier behind it, the MFMA segment and the wait behind that - for the LDS-resident input
patch (default) and the gathered form (PV_CONV_PATCH=0), on the 64 x 64 and 32 x 32 shapes.   usage (GPU box): python tools/diag/convbig_seg_stamps.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r"""
import ctypes, os, sys, torch
sys.path.insert(0, %r)
from photoverse_amd import _lib
_lib.LIB = "/tmp/libpv_convbig_stamps.so"
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, cin, cout, h, bm = [int(v) for v in sys.argv[1:6]]
torch.manual_seed(0)
x = torch.randn(B * h * h, cin, device=dev).half()
w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
rec = Recorder(dev)
rec.big_min = bm
rec.gemm(x, w, bias=torch.zeros(cout, device=dev), conv=dict(batch=B, hin=h, win=h, hout=h, wout=h), colstats=True)
for _ in range(300):
    rec.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    rec.run()
e1.record()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * (3 * 8 * 8))()
fn = rec.lib.pv_convbig_read_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
print("%%s  PV_CONV_PATCH=%%s: %%s, launch %%.1f us (stamped build)" %% (sys.argv[1:6], os.environ.get("PV_CONV_PATCH", "1"), rec.tags[-1][0], e0.elapsed_time(e1) / 50 * 1e3))
print("  wg wave |   LOAD  wait-1   MFMA  wait-2 | per stage | stages | clock GHz")
for wg in range(3):
    for wv in range(8):
        v = [out[(wg * 8 + wv) * 8 + k] for k in range(8)]
        n = max(v[6], 1)
        per = [x_ / n for x_ in v[:4]]
        print("  %%2d  %%d   | %%6.0f %%6.0f %%6.0f %%6.0f |  %%7.0f  |  %%4d  |  %%.2f" %% (wg, wv, per[0], per[1], per[2], per[3], sum(per), n, v[4] / max(v[5], 1) * 0.1))
""" % ROOT


def main():
    import photoverse_amd.build as b
    objs = []
    for f in b.SOURCES:
        o = "/tmp/cbst_%s.o" % f
        extra = (["-DPV_CONVBIG_STAMPS"] + os.environ.get("CB_FLAGS", "").split()) if f == "pv_convbig.hip" else []
        subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), *extra, "-c", os.path.join(b.CSRC, f), "-o", o], stderr=subprocess.DEVNULL)
        objs.append(o)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "/tmp/libpv_convbig_stamps.so", *objs])
    for shape in (("16", "320", "320", "64", "256"), ("16", "640", "640", "32", "128")):
        for patch in ("1", "0"):
            r = subprocess.run([sys.executable, "-c", CHILD, *shape], env=dict(os.environ, PV_CONV_PATCH=patch), capture_output=True, text=True, timeout=600)
            print(r.stdout + (r.stderr[-3000:] if r.returncode else ""), flush=True)


if __name__ == "__main__":
    main()
