#!/usr/bin/env python3
"""Diagnostic: shader-clock stamps of one workgroup of pv_xq.hip's head-parallel attn2 kernel (stamped COPY, private library):
entry | K/V DMA issue + prologue | GEMM loop | norm2 fold + SDPA + stores.  B = 16, nq = 256 (16 x 16 level), C = 1280."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402
s = open(os.path.join(b.CSRC, "pv_xq.hip")).read()
BLK = os.environ.get("GC_BLOCK", "100")
s = s.replace('#include "pv_common.h"', '#include "%s"\n__device__ unsigned long long gc_stamps[128];\n'
              '#define STAMP(i) do { if (blockIdx.x == %s && (threadIdx.x & 63) == 0) gc_stamps[(i) + 16 * (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); } while (0)'
              % (os.path.join(b.CSRC, "pv_common.h"), BLK), 1)


def sub1(s, old, new):
    assert s.count(old) == 1, (old, s.count(old))
    return s.replace(old, new, 1)


s = sub1(s, "    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();\n    const int fr = lane & 15, fq = lane >> 4;\n    const int C = p.heads * p.d;",
         "    STAMP(0);\n    const int tid = threadIdx.x, lane = tid & 63, wave = pv_wave_id();\n    const int fr = lane & 15, fq = lane >> 4;\n    const int C = p.heads * p.d;")
s = sub1(s, "    // ---- GEMM: Q^T = Wq'[head rows] . X^T on the raw rows ----", "    STAMP(1);\n    // ---- GEMM: Q^T = Wq'[head rows] . X^T on the raw rows ----")
# the computing waves' loop (the loader waves have left through their own branch before it)
s = sub1(s, "    // computing waves: no vector-memory operation in the loop; behind the barrier stage kt has landed and stage kt-1 (refilled next) is free\n    for (int kt = 0; kt < nk; ++kt) {",
         "    STAMP(2);\n    for (int kt = 0; kt < nk; ++kt) {\n        if (kt == 9) STAMP(6);\n        if (kt == 12) STAMP(7);")
s = sub1(s, "    // ---- norm2 fold, query bias, B operand of the score product ----", "    STAMP(3);\n    // ---- norm2 fold, query bias, B operand of the score product ----")
s = sub1(s, "    // ---- per head: S^T = K . Q^T, two softmaxes", "    STAMP(4);\n    // ---- per head: S^T = K . Q^T, two softmaxes")
s = s.rstrip()
tail = "}\n\n}  // namespace"
i = s.index(tail)
s = s[:i] + "    STAMP(5);\n" + s[i:]
s += '\nextern "C" int pv_gc_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gc_stamps), 128 * 8); }\n'
src, lib = "/tmp/pv_xq_stamps.hip", "/tmp/libpv_diag_xq.so"
open(src, "w").write(s)
if os.environ.get("GC_DRY"):
    sys.exit(0)
objs = []
for f in b.SOURCES:
    o = os.path.join(b.LIBDIR, f.replace(".hip", ".o"))
    if f == "pv_xq.hip":
        o = "/tmp/diag_xq.o"
        subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), "-I", b.CSRC, "-c", src, "-o", o])
    objs.append(o)
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
from photoverse_amd import _lib  # noqa: E402
_lib.LIB = lib
from photoverse_amd.ops import Recorder  # noqa: E402
dev = torch.device("cuda")
B, n, C, p = 16, int(os.environ.get("GC_N", "256")), 1280, 1
h16 = lambda *s_, scale=1.0: (torch.randn(*s_, device=dev) * scale).half()
hs, kvt, kvi = h16(B * n, C), h16(B * 77, 2 * C), h16(B * p, 2 * C)
wq = h16(C, C, scale=0.03)
g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rec = Recorder(dev)
rec.cross_attention_lnq(hs, wq, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, nq=n, nt=77, nip=p, ln_gamma=g, ln_beta=bt)
for _ in range(50):
    rec.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    rec.run()
e1.record()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 128)()
fn = rec.lib.pv_gc_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
t = list(out)
print(f"launch {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; workgroup {BLK}; shader cycles per wave:")
for w in range(4):                      # the computing waves (waves 4-7 are the loaders: they return from inside the GEMM section)
    u = t[16 * w:16 * w + 16]
    print(f"  wave {w}: entry {u[1] - u[0]:6d} | setup {u[2] - u[1]:6d} | GEMM loop ({C // 64} steps) {u[3] - u[2]:6d} (steps 9-11: {u[7] - u[6]}) | "
          f"norm2 fold {u[4] - u[3]:6d} | SDPA + stores {u[5] - u[4]:6d} | total {u[5] - u[0]:6d}")
