"""A/B of one -D macro on pv_attnbwd.hip (two private libraries under /tmp), backward d = 40, B = 16, N = 4096, sustained.  usage: ... <MACRO>"""
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
outs = rec.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, batch=B, heads=H, nq=n, nk=n, d=d)
for _ in range(250): rec.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(150): rec.run()
e1.record(); torch.cuda.synchronize()
import hashlib
print("%%.1f us  sha %%s" %% (e0.elapsed_time(e1) / 150 * 1e3, hashlib.sha1(torch.cat([t.float().flatten() for t in outs]).cpu().numpy().tobytes()).hexdigest()[:10]))
""" % ROOT
import photoverse_amd.build as b
macro = sys.argv[1]
base = [os.path.join(b.LIBDIR, f.replace(".hip", ".o")) for f in b.SOURCES if f != "pv_attnbwd.hip"]
libs = []
for name, ex in (("base", []), ("macro", ["-D" + macro])):
    o = "/tmp/bm_%s.o" % name
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get("pv_attnbwd.hip", []), *ex, "-c", os.path.join(b.CSRC, "pv_attnbwd.hip"), "-o", o], stderr=subprocess.DEVNULL)
    lib = "/tmp/libpv_bm_%s.so" % name
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o, *base])
    libs.append((name, lib))
for r in range(3):
    for name, lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        print("round %d  %-5s %s" % (r, name, out.stdout.strip() or "ERR " + out.stderr[-300:]), flush=True)
