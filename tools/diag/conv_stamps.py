#!/usr/bin/env python3
"""Diagnostic: shader-clock stamps of one workgroup of the 3x3 conv kernel (stamped COPY of pv_gemm.hip, private library):
kernel entry | staging geometry + first stages issued and landed | main loop | epilogue.  conv 320 -> 320 @ 64x64, B = 16, with column statistics."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402
s = open(os.path.join(b.CSRC, "pv_gemm.hip")).read()
s = s.replace('#include "pv_common.h"', '#include "%s"\n__device__ unsigned long long gc_stamps[16];\n'
              '#define STAMP(i) do { if (blockIdx.x == %s && threadIdx.x == 0) gc_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)'
              % (os.path.join(b.CSRC, "pv_common.h"), os.environ.get("GC_BLOCK", "300")), 1)
marks = [("    const int lane = pv_lane_id();\n    const int wave = pv_wave_id();\n    // tpw (tiles per workgroup)", 0),
         ("    half8_t xa0[MI], wb0[NF], xa1[MI], wb1[NF];\n    if (T > 0) read_half(xa0, wb0, 0, 0);", 1),
         ("    // ---- epilogue ---------------------------------------------------------------------------\n    if (gridDim.y > 1) {", 2),
         ("    // ---- next tile of the group ----", 3)]
for m, i in marks:
    assert s.count(m) == 1, (i, s.count(m))
    s = s.replace(m, "    STAMP(%d);\n" % i + m, 1)
s += '\nextern "C" int pv_gc_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gc_stamps), 16 * 8); }\n'
src, lib = "/tmp/pv_gemm_stamps.hip", "/tmp/libpv_diag_gemm.so"
open(src, "w").write(s)
objs = []
for f in b.SOURCES:
    path = src if f == "pv_gemm.hip" else os.path.join(b.CSRC, f)
    o = f"/tmp/diagg_{f}.o"
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), "-I", b.CSRC, "-c", path, "-o", o])
    objs.append(o)
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
from photoverse_amd import _lib  # noqa: E402
_lib.LIB = lib
from photoverse_amd.ops import Recorder  # noqa: E402
dev = torch.device("cuda")
B, hw, cin, cout = 16, 64, 320, 320
x = (torch.randn(B * hw * hw, cin, device=dev)).half()
w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
rec = Recorder(dev)
cs = os.environ.get("GC_COLSTATS", "1") == "1"
rec.gemm(x, w, bias=torch.zeros(cout, device=dev), conv=dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw), colstats=cs)
for _ in range(3):
    rec.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    rec.run()
e1.record()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
fn = rec.lib.pv_gc_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
t = list(out)[:4]
print(f"launch {e0.elapsed_time(e1) / 5 * 1e3:.1f} us (colstats={cs}); workgroup {os.environ.get('GC_BLOCK', '300')}, wave 0, shader cycles:")
print(f"  geometry + first stages issued / landed  {t[1] - t[0]:7d}")
print(f"  main loop (45 K-steps)                   {t[2] - t[1]:7d}")
print(f"  epilogue                                 {t[3] - t[2]:7d}")
