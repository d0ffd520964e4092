#!/usr/bin/env python3
"""Diagnostic: in-kernel s_memtime stamps of one wave of the d=40 self-attention kernel (stamped COPY of pv_attn.hip, private lib).
Per 64-key tile of the LDS-DMA kernel: wait + barrier + DMA issue (even tiles) | QK^T MFMAs | softmax | P.V MFMAs.  XA_ABLATE=nobar drops the
two workgroup barriers per tile (wrong results, timing only)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402
ABL = os.environ.get("XA_ABLATE", "")
s = open(os.path.join(b.CSRC, "pv_attn.hip")).read()
s = s.replace('#include "pv_common.h"', '#include "%s"\n__device__ unsigned long long at_stamps[16];\n'
              '#define STAMP(i) do { if (D == 40 && blockIdx.x == 1000 && threadIdx.x == 0 && (t == 20 || t == 21)) at_stamps[i + 8 * (t - 20)] = __builtin_amdgcn_s_memtime(); } while (0)'
              % os.path.join(b.CSRC, "pv_common.h"))
rep = [("            if ((t & 1) == 0) {\n                // ONE barrier per PAIR of tiles", "            STAMP(0);\n            if ((t & 1) == 0) {\n                // ONE barrier per PAIR of tiles"),
       ("            const bool need_mask = p.causal || (t + 1) * KB > p.nk;\n            tile(t, t & 3, need_mask, t == 0);\n", "            STAMP(1);\n            const bool need_mask = p.causal || (t + 1) * KB > p.nk;\n            tile(t, t & 3, need_mask, t == 0);\n            STAMP(5);\n"),
       ("        half8_t pb[2][NQ];\n#pragma unroll\n        for (int qi = 0; qi < NQ; ++qi) {\n            if (MASKED) {", "        STAMP(3);\n        half8_t pb[2][NQ];\n#pragma unroll\n        for (int qi = 0; qi < NQ; ++qi) {\n            if (MASKED) {"),
       ("#pragma unroll\n        for (int s2 = 0; s2 < 2; ++s2)\n#pragma unroll\n            for (int f = 0; f < C::DVF; ++f) {\n                const half8_t a = vt_frag(sV, C::VS, s2 * 32, f * 16, fr, fq);",
        "        STAMP(4);\n#pragma unroll\n        for (int s2 = 0; s2 < 2; ++s2)\n#pragma unroll\n            for (int f = 0; f < C::DVF; ++f) {\n                const half8_t a = vt_frag(sV, C::VS, s2 * 32, f * 16, fr, fq);")]
for a, c in rep:
    assert a in s, a[:60]
    s = s.replace(a, c, 1)
if ABL == "nobar":      # timing only (wrong results): the DMA ring without its workgroup barrier
    a = '                __builtin_amdgcn_s_barrier();\n                asm volatile("" ::: "memory");\n                if (t + 2 < ntiles) issue_tile(t + 2);'
    assert a in s
    s = s.replace(a, a.replace("__builtin_amdgcn_s_barrier();", ""), 1)
s += '\nextern "C" int pv_at_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(at_stamps), 16 * 8); }\n'
src, lib = "/tmp/pv_attn_stamps.hip", "/tmp/libpv_diag_attn.so"
open(src, "w").write(s)
objs = []
for f in b.SOURCES:
    path = src if f == "pv_attn.hip" else os.path.join(b.CSRC, f)
    o = f"/tmp/diaga_{f}.o"
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), "-I", b.CSRC, "-c", path, "-o", o])
    objs.append(o)
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
from photoverse_amd import _lib  # noqa: E402
_lib.LIB = lib
from photoverse_amd.ops import Recorder  # noqa: E402
dev = torch.device("cuda")
B, n, d = 16, 4096, 40
C = 8 * d
qkv = (torch.randn(B * n, 3 * C, device=dev) * 0.5).half()
rec = Recorder(dev)
rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=8, nq=n, nk=n, d=d)
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
fn = rec.lib.pv_at_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
names = ["wait + barrier + DMA issue (even tile)", "-", "QK^T (32 MFMA + reads)", "softmax (VALU)", "P.V (24 MFMA + tr reads)"]
print(f"ablate={ABL or '-'}  launch {e0.elapsed_time(e1) / 5 * 1e3:.1f} us;  tiles 20 / 21 of workgroup 1000, wave 0:")
for k in range(2):
    t = list(out)[8 * k:8 * k + 6]
    t = [t[0], t[1], t[1], t[3], t[4], t[5]]
    for i in range(5):
        print(f"  tile {20 + k}: {names[i]:34s} {t[i + 1] - t[i]:6d} cycles")
    print(f"  tile {20 + k}: {'total':34s} {t[5] - t[0]:6d} cycles")
