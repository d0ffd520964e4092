import os, subprocess, sys, concurrent.futures as cf
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
print("%%.1f" %% (e0.elapsed_time(e1) / 400 * 1e3))
""" % ROOT
import photoverse_amd.build as b
pads = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else [str(i) for i in range(16)])]
base = []
for f in b.SOURCES:
    if f != "pv_attn.hip":
        base.append(os.path.join(b.LIBDIR, f.replace(".hip", ".o")))      # the in-tree objects travel with the snapshot
def build(pad):
    o = "/tmp/pad_%d.o" % pad
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get("pv_attn.hip", []), "-DPV_ATTN8_LOOP_PAD=%d" % pad, "-c", os.path.join(b.CSRC, "pv_attn.hip"), "-o", o],
                          stderr=subprocess.DEVNULL)
    lib = "/tmp/libpv_pad%d.so" % pad
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o, *base])
    return lib
with cf.ThreadPoolExecutor(8) as ex:
    libs = dict(zip(pads, ex.map(build, pads)))
for r in range(3):
    row = []
    for pad in pads:
        out = subprocess.run([sys.executable, "-c", CHILD, libs[pad]], capture_output=True, text=True)
        row.append(out.stdout.strip() or "ERR " + out.stderr[-200:])
    print("round %d  " % r + "  ".join("pad%d %s" % (p, v) for p, v in zip(pads, row)), flush=True)
