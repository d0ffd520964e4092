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
B, n, H, d = 16, 4096, 8, 40
C = H * d
g = torch.Generator().manual_seed(7)
qkv = torch.randn(B * n, 3 * C, generator=g).half().cuda()
do = torch.randn(B * n, C, generator=g).half().cuda()
pre = Recorder(dev)
lse = torch.empty((B, H, n), dtype=torch.float32, device=dev)
o = pre.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=H, nq=n, nk=n, d=d, lse=lse)
pre.run()
rec = Recorder(dev)
rec.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, batch=B, heads=H, nq=n, nk=n, d=d)
for _ in range(250): rec.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(150): rec.run()
e1.record(); torch.cuda.synchronize()
print("%%.1f" %% (e0.elapsed_time(e1) / 150 * 1e3))
""" % ROOT
import photoverse_amd.build as b
base = [os.path.join(b.LIBDIR, f.replace(".hip", ".o")) for f in b.SOURCES if f != "pv_attnbwd.hip"]
cfgs = [("none", -1, -1)] + [("kv%d" % k, k, -1) for k in range(8)] + [("q%d" % k, -1, k) for k in range(8)]
def build(c):
    name, kv, q = c
    o = "/tmp/bpad_%s.o" % name
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get("pv_attnbwd.hip", []), "-DPV_ATTN8_BWD_PAD_KV=%d" % kv, "-DPV_ATTN8_BWD_PAD_Q=%d" % q, "-c",
                           os.path.join(b.CSRC, "pv_attnbwd.hip"), "-o", o], stderr=subprocess.DEVNULL)
    lib = "/tmp/libpv_bpad_%s.so" % name
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o, *base])
    return lib
with cf.ThreadPoolExecutor(8) as ex:
    libs = list(ex.map(build, cfgs))
for r in range(2):
    row = []
    for c, lib in zip(cfgs, libs):
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        row.append("%s %s" % (c[0], out.stdout.strip() or "ERR " + out.stderr[-200:]))
    print("round %d  " % r + "  ".join(row), flush=True)
