#!/usr/bin/env python3
"""Diagnostic: shader-clock stamps of one workgroup of pv_convbig.hip's 256 x 320 kernel (stamped COPY, private library):
entry | prologue | main loop | epilogue, the eight phases of one 64-deep K-step (GC_STEP, default 20) with the wait + barrier of its two stages,
and the in-kernel clock (s_memtime / s_memrealtime).  conv 320 -> 320 @ 64x64, B = 16."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402
s = open(os.path.join(b.CSRC, "pv_convbig.hip")).read()
STEP = os.environ.get("GC_STEP", "20")
s = s.replace('#include "pv_gemm_dev.h"', '#include "%s"\n__device__ unsigned long long gc_stamps[64];\n'
              '#define STAMP(i) do { if (blockIdx.x == %s && (threadIdx.x & 255) == 0) gc_stamps[(i) + 32 * (threadIdx.x >> 8)] = __builtin_amdgcn_s_memtime(); } while (0)\n'
              '#define RSTAMP(i) do { if (blockIdx.x == %s && threadIdx.x == 0) gc_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)'
              % (os.path.join(b.CSRC, "pv_gemm_dev.h"), os.environ.get("GC_BLOCK", "100"), os.environ.get("GC_BLOCK", "100")), 1)
def sub1(s, old, new):
    assert s.count(old) == 1, (old, s.count(old))
    return s.replace(old, new, 1)


s = sub1(s, "    const int lane = pv_lane_id();\n    const int wave = pv_wave_id();\n    const int bid = pv_xcd_remap", "    STAMP(0); RSTAMP(30);\n    const int lane = pv_lane_id();\n    const int wave = pv_wave_id();\n    const int bid = pv_xcd_remap")
s = sub1(s, "    KPos kn = kpos_of(3);", "    STAMP(1);\n    KPos kn = kpos_of(3);")
s = sub1(s, "        read_frags(s);\n", "        if (s == %s) STAMP(4);\n        read_frags(s);\n" % STEP)
s = sub1(s, "    // ---- epilogue (pv_gemm.hip's", "    STAMP(2); RSTAMP(31);\n    // ---- epilogue (pv_gemm.hip's")
# the four segment barriers of the loop body: stamp in front of and behind each
body0, body1 = s.index("        if (s == %s) STAMP(4);" % STEP), s.index("    if (wm == 0) seg_barrier();")
body = s[body0:body1]
assert body.count("        seg_barrier();\n") == 2
parts = body.split("        seg_barrier();\n")
out = parts[0]
for i in range(2):
    out += "        if (s == %s) STAMP(%d);\n        seg_barrier();\n        if (s == %s) STAMP(%d);\n" % (STEP, 5 + 2 * i, STEP, 6 + 2 * i) + parts[i + 1]
s = s[:body0] + out + s[body1:]
# end of kernel: behind the group epilogue's loop
tail = "        for (int hb = 0; hb < MI / 4; ++hb) block(hb);\n    }\n}"
s = sub1(s, tail, tail[:-1] + "    STAMP(3);\n}")
s += '\nextern "C" int pv_gc_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gc_stamps), 64 * 8); }\n'
src, lib = "/tmp/pv_convbig_stamps.hip", "/tmp/libpv_diag_convbig.so"
open(src, "w").write(s)
objs = []
for f in b.SOURCES:
    path = src if f == "pv_convbig.hip" else os.path.join(b.CSRC, f)
    o = f"/tmp/diagcb_{f}.o"
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
assert rec.tags[-1][0].startswith("big_tile_kernel"), rec.tags[-1]
for _ in range(200):
    rec.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    rec.run()
e1.record()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
fn = rec.lib.pv_gc_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
t = list(out)
clk = (t[2] - t[0]) / max(t[31] - t[30], 1) * 100.0
print(f"launch {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (colstats={cs}); workgroup {os.environ.get('GC_BLOCK', '100')}; in-kernel clock {clk:.0f} MHz; shader cycles:")
for half, label in ((0, "wave 0"), (32, "wave 4")):
    u = t[half:half + 32]
    print(f" {label}: prologue {u[1] - u[0]:6d} | main loop (90 stages) {u[2] - u[1]:7d} = {(u[2] - u[1]) / 90:.0f} per stage | epilogue {u[3] - u[2]:6d}")
    print(f"   stage {STEP}: LOAD {u[5] - u[4]} (+ barrier wait {u[6] - u[5]}) | MFMA {u[7] - u[6]} (+{u[8] - u[7]}) = {u[8] - u[4]} per 32-deep stage")
