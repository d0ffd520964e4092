"""Loop-alignment sweep for pv_convbig.hip (PV_CONVBIG_PAD_GATHER / PV_CONVBIG_PAD_PATCH): eight builds per macro, four shapes timed (sustained)."""
import os, subprocess, sys, concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from photoverse_amd import _lib
_lib.LIB = sys.argv[1]
from photoverse_amd.ops import Recorder, pack_geglu
dev = torch.device("cuda")
B = 16
torch.manual_seed(0)
def h16(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).half()
def timeit(rec):
    rec.run(); torch.cuda.synchronize()
    for _ in range(1500): rec.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(600): rec.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 600 * 1e3
def conv(cin, cout, hw, cs=True, c1=0):
    rec = Recorder(dev)
    x = h16(B * hw * hw, cin); x1 = h16(B * hw * hw, c1) if c1 else None
    w = h16(cout, 9 * (cin + c1), scale=0.02)
    rec.gemm(x, w, a1=x1, bias=torch.zeros(cout, device=dev), conv=dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw, stride=1, upsample=0), colstats=cs)
    return timeit(rec)
def gemm(M, K, N, geglu=False):
    rec = Recorder(dev)
    x, w = h16(M, K), h16(N, K, scale=0.02); b = torch.zeros(N, device=dev)
    if geglu: w, b = pack_geglu(w, b)
    r = None if geglu else h16(M, N)
    rec.gemm(x, w, bias=b, residual=r, geglu=geglu)
    return timeit(rec)
out = []
which = sys.argv[2]
if which == "patch":
    out.append("c320@64 %%.1f" %% conv(320, 320, 64)); out.append("c640>320@64 %%.1f" %% conv(320, 320, 64, c1=320))
else:
    out.append("c640@32 %%.1f" %% conv(640, 640, 32)); out.append("ff2@64 %%.1f" %% gemm(65536, 1280, 320)); out.append("geglu@32 %%.1f" %% gemm(16384, 640, 5120, geglu=True))
    out.append("c1280>640@32 %%.1f" %% conv(640, 640, 32, c1=640))
print("  ".join(out))
""" % ROOT
import photoverse_amd.build as b
base = [os.path.join(b.LIBDIR, f.replace(".hip", ".o")) for f in b.SOURCES if f != "pv_convbig.hip"]
which = sys.argv[1]
macro = "PV_CONVBIG_PAD_PATCH" if which == "patch" else "PV_CONVBIG_PAD_GATHER"
pads = [-1] + list(range(8))
def build(pad):
    o = "/tmp/cpad_%s_%d.o" % (which, pad)
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get("pv_convbig.hip", []), "-D%s=%d" % (macro, pad), "-c", os.path.join(b.CSRC, "pv_convbig.hip"), "-o", o],
                          stderr=subprocess.DEVNULL)
    lib = "/tmp/libpv_cpad_%s_%d.so" % (which, pad)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o, *base])
    return lib
with cf.ThreadPoolExecutor(9) as ex:
    libs = list(ex.map(build, pads))
for r in range(2):
    for pad, lib in zip(pads, libs):
        out = subprocess.run([sys.executable, "-c", CHILD, lib, which], capture_output=True, text=True)
        print("round %d  %s pad %2d   %s" % (r, which, pad, out.stdout.strip() or "ERR " + out.stderr[-300:]), flush=True)
