"""Loop-alignment sweep (-DPV_LOOP_PAD_TEST=k) for row_gemm_kernel (pv_rowgemm.hip) and gemm_conv_kernel (pv_gemm.hip)."""
import os, subprocess, sys, concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r"""
import os, sys, torch
sys.path.insert(0, %r)
from photoverse_amd import _lib
_lib.LIB = sys.argv[1]
from photoverse_amd.ops import Recorder, pack_geglu, pack_geglu_rows
dev = torch.device("cuda")
B = 16
torch.manual_seed(0)
def h16(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).half()
def timeit(rec):
    rec.run(); torch.cuda.synchronize()
    for _ in range(2000): rec.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(800): rec.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 800 * 1e3
out = []
if sys.argv[2] == "row":
    M, C = 65536, 320
    x, g, bt = h16(M, C), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w, b = h16(2560, C, scale=0.05), torch.zeros(2560, device=dev)
    wp, bp = pack_geglu_rows(w, b)
    rec = Recorder(dev); rec.row_gemm(x, wp, bias=bp, ln_gamma=g, ln_beta=bt, geglu=True); out.append("ln+ff1 geglu %%.1f" %% timeit(rec))
    w3 = h16(960, C, scale=0.05)
    rec = Recorder(dev); rec.row_gemm(x, w3, bias=torch.zeros(960, device=dev), ln_gamma=g, ln_beta=bt); out.append("ln+qkv %%.1f" %% timeit(rec))
else:
    def gemm(M, K, N, res=True):
        rec = Recorder(dev); x, w = h16(M, K), h16(N, K, scale=0.02)
        rec.gemm(x, w, bias=torch.zeros(N, device=dev), residual=h16(M, N) if res else None); return timeit(rec)
    def conv(cin, cout, hw):
        rec = Recorder(dev); x = h16(B * hw * hw, cin); w = h16(cout, 9 * cin, scale=0.02)
        rec.gemm(x, w, bias=torch.zeros(cout, device=dev), conv=dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw, stride=1, upsample=0), colstats=True); return timeit(rec)
    out.append("to_out 320@64 %%.1f" %% gemm(65536, 320, 320)); out.append("ff2 2560>640@32 %%.1f" %% gemm(16384, 2560, 640)); out.append("conv1280@16 %%.1f" %% conv(1280, 1280, 16))
    out.append("640>640@32 %%.1f" %% gemm(16384, 640, 640))
print("  ".join(out))
""" % ROOT
import photoverse_amd.build as b
which = sys.argv[1]
src = "pv_rowgemm.hip" if which == "row" else "pv_gemm.hip"
base = [os.path.join(b.LIBDIR, f.replace(".hip", ".o")) for f in b.SOURCES if f != src]
pads = [-1] + list(range(8))
def build(pad):
    o = "/tmp/gpad_%s_%d.o" % (which, pad)
    ex = [] if pad < 0 else ["-DPV_LOOP_PAD_TEST=%d" % pad]
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *ex, "-c", os.path.join(b.CSRC, src), "-o", o], stderr=subprocess.DEVNULL)
    lib = "/tmp/libpv_gpad_%s_%d.so" % (which, pad)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o, *base])
    return lib
with cf.ThreadPoolExecutor(9) as ex:
    libs = list(ex.map(build, pads))
for r in range(2):
    for pad, lib in zip(pads, libs):
        out = subprocess.run([sys.executable, "-c", CHILD, lib, which], capture_output=True, text=True)
        print("round %d  %s pad %2d   %s" % (r, which, pad, out.stdout.strip() or "ERR " + out.stderr[-400:]), flush=True)
