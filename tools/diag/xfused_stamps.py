#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock stamps of one mid-launch workgroup of the fused attn2 kernel.  Builds a STAMPED COPY of
photoverse_amd/csrc/pv_xfused.hip into a private library (the product kernel carries no stamps)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402


ABL = os.environ.get("XF_ABLATE", "")     # timing ablations (WRONG results): nobar


def stamped_source():
    s = open(os.path.join(b.CSRC, "pv_xfused.hip")).read()
    if "nobar" in ABL:      # no workgroup barriers at all
        s = s.replace("        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n    };", "        asm volatile(\"\" ::: \"memory\");\n    };")
    s = s.replace('#include "pv_common.h"', '#include "%s"\n__device__ unsigned long long xf_stamps[16];\n'
                  '#define STAMP(i) do { if (blockIdx.x == 100 && threadIdx.x == 0) xf_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)'
                  % os.path.join(b.CSRC, "pv_common.h"))
    marks = ["    half8_t xf[KK][NQ];\n    int mrow[NQ];", "    if (!SPAN) issue_group(0, buf1);", "    // The register loads above made the compiler wait;",
             "    // ---- phase 1: Q^T = Wq", "    // ---- phase 2: dual-branch attention, one", "    // ---- phase 3: out^T = Wo'"]
    for i, m in enumerate(marks):
        assert m in s, m
        s = s.replace(m, "STAMP(%d);\n" % i + m, 1)
    idx = s.index('}  // namespace\n\nextern "C" int pv_xattn_pack_kv')
    k = s.rfind("}\n", 0, idx)
    s = s[:k] + "STAMP(6);\n" + s[k:]
    s += '\nextern "C" int pv_xf_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xf_stamps), 16 * 8); }\n'
    return s


src = "/tmp/pv_xfused_stamps.hip"
open(src, "w").write(stamped_source())
lib = "/tmp/libpv_diag.so"
objs = []
for s in b.SOURCES:       # the other sources: the in-tree objects (they travel with the snapshot)
    if s == "pv_xfused.hip":
        o = "/tmp/diag_pv_xfused.o"
        subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(s, []), "-I", b.CSRC, "-c", src, "-o", o], stderr=subprocess.DEVNULL)
    else:
        o = os.path.join(b.LIBDIR, s.replace(".hip", ".o"))
    objs.append(o)
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
from photoverse_amd import _lib  # noqa: E402
_lib.LIB = lib
from photoverse_amd.ops import Recorder  # noqa: E402

dev = torch.device("cuda")
C = int(os.environ.get("XF_C", "320"))          # 320 (n = 4096) or 640 (n = 1024)
B, n, d, p = 16, 4096 * 320 * 320 // (C * C), C // 8, 1
h16 = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).half()
hs, kvt, kvi = h16(B * n, C), h16(B * 77, 2 * C), h16(B * p, 2 * C)
wq, wo, bo = h16(C, C, scale=0.05), h16(C, C, scale=0.05), torch.zeros(C, device=dev)
g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rec = Recorder(dev)
kimg, vimg = rec.xattn_pack_kv(kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, d=d, nt=77, nip=p)
rec.cross_attention_fused(hs, wq, rec.pack_wo_for_fused(wo), bo, kimg, vimg, batch=B, nq=n, heads=8, d=d, nt=77, nip=p, ln_gamma=g, ln_beta=bt)
for _ in range(5):
    rec.run()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
fn = rec.lib.pv_xf_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
names = ["", "issue X loads", "issue DMA + LayerNorm (waits for X)", "drain", "phase 1 (to_q)", "phase 2 (attention)", "phase 3 (to_out) + epilogue"]
t = list(out)[:7]
for i in range(1, 7):
    print(f"{names[i]:40s} {t[i] - t[i - 1]:8d} shader cycles")
print(f"{'total (wave 0 of workgroup 100)':40s} {t[6] - t[0]:8d} shader cycles (s_memtime)")
