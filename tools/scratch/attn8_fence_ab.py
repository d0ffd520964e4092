import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from photoverse_amd import _lib
_lib.LIB = sys.argv[1]
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, n, d = 16, 4096, 40
C = 8 * d
torch.manual_seed(0)
qkv = torch.randn(B * n, 3 * C, device=dev).half()
rec = Recorder(dev)
rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=8, nq=n, nk=n, d=d)
for _ in range(800): rec.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(400): rec.run()
e1.record(); torch.cuda.synchronize()
print("%%s  %%.1f us" %% (sys.argv[1], e0.elapsed_time(e1) / 400 * 1e3))
""" % ROOT
import photoverse_amd.build as b
libs = []
MACRO = sys.argv[1] if len(sys.argv) > 1 else "PV_ATTN8_NO_PRO_FENCE"
for name, extra in (("fence", []), ("nofence", ["-D" + MACRO])):
    objs = []
    for f in b.SOURCES:
        o = "/tmp/fab_%s_%s.o" % (name, f)
        ex = extra if f == "pv_attn.hip" else []
        if f == "pv_attn.hip" or name == "fence":
            subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), *ex, "-c", os.path.join(b.CSRC, f), "-o", o], stderr=subprocess.DEVNULL)
        else:
            o = "/tmp/fab_fence_%s.o" % f
        objs.append(o)
    lib = "/tmp/libpv_%s.so" % name
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    libs.append(lib)
for r in range(4):
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        print("round %d  %s" % (r, out.stdout.strip() or out.stderr[-300:]), flush=True)
