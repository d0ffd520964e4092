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


ABL = os.environ.get("XF_ABLATE", "")     # timing ablations (WRONG results): nodma | nomfma | nobar | nostamp-fine


def stamped_source():
    s = open(os.path.join(b.CSRC, "pv_xfused.hip")).read()
    if "nodma" in ABL:      # ring stages are never issued (phase 1 / 3 read stale LDS)
        s = s.replace("        if (wave >= 4) return;                                // waves 0-3 are the DMA issuers (5 pieces each), see below",
                      "        return;")
    if "nomfma" in ABL:     # phase 1 / 3 MFMAs dropped (operands kept alive)
        s = s.replace("acc[i][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[i], xf[2 * kt][qi], acc[i][qi], 0, 0, 0);",
                      'asm volatile("" :: "v"(a0[i]), "v"(xf[2 * kt][qi]));')
        s = s.replace("acc[i][qi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], xf[2 * kt + 1][qi], acc[i][qi], 0, 0, 0);",
                      'asm volatile("" :: "v"(a1[i]), "v"(xf[2 * kt + 1][qi]));')
    if "nobar" in ABL:      # no workgroup barriers at all
        s = s.replace("        __builtin_amdgcn_s_barrier();\n        asm volatile(\"\" ::: \"memory\");\n    };", "        asm volatile(\"\" ::: \"memory\");\n    };")
    if "noread" in ABL:     # phase 1 / 3 fragment reads dropped
        s = s.replace("        for (int i = 0; i < 5; ++i) a[i] = ld_frag128(sw, i * 16 + fr, ks * 4 + g);",
                      '        for (int i = 0; i < 5; ++i) { a[i] = half8_t{(half_t)lane}; asm volatile("" : "+v"(a[i])); }')
    s = s.replace('#include "pv_common.h"', '#include "%s"\n__device__ unsigned long long xf_stamps[16];\n'
                  '#define STAMP(i) do { if (blockIdx.x == 300 && threadIdx.x == 0) xf_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)'
                  % os.path.join(b.CSRC, "pv_common.h"))
    marks = ["    half8_t xf[KK][2];\n    int mrow[2];", "    issue_kv(regB, 0);\n", "    // The register loads above made the compiler wait;",
             "    // ---- phase 1: Q^T = Wq", "    // ---- phase 2: dual-branch attention; this", "    // ---- exchange: every wave needs",
             "    // ---- phase 3: out^T = Wo'"]
    for i, m in enumerate(marks):
        assert m in s, m
        s = s.replace(m, "STAMP(%d);\n" % i + m, 1)
    # fine stamps inside ring stage t = 5 of phase 1
    fine = [("            read_half(a1, regA, t, 1);\n", "            if (t == 5) STAMP(8);\n", True),
            ("                wait_stages((t + S - 1 < NT ? t + S - 1 : NT - 1) - (t + 1));   // stage t+1 landed", "                if (t == 5) STAMP(9);\n", True),
            ("                if (t + S < NT) issue_stage(rq, regA, t + S);\n", "                if (t == 5) STAMP(10);\n", True),
            ("                read_half(a0, regA, t + 1, 0);\n", "                if (t == 5) STAMP(11);\n", True),
            ("            if (kt == KT - 1) {\n#pragma unroll\n                for (int i = 0; i < 5; ++i)\n#pragma unroll\n                    for (int qi = 0; qi < 2; ++qi)\n#pragma unroll\n                        for (int r = 0; r < 4; ++r) qf[", "            if (t == 5) STAMP(12);\n", True)]
    for m, ins, before in fine:
        if "nofine" in ABL:
            break
        assert m in s, m
        s = s.replace(m, ins + m, 1)
    idx = s.index('}  // namespace\n\nextern "C" int pv_xattn_pack_kv')
    k = s.rfind("}\n", 0, idx)
    s = s[:k] + "STAMP(7);\n" + s[k:]
    s += '\nextern "C" int pv_xf_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xf_stamps), 16 * 8); }\n'
    return s


src = "/tmp/pv_xfused_stamps.hip"
open(src, "w").write(stamped_source())
lib = "/tmp/libpv_diag.so"
objs = []
for s in b.SOURCES:
    path = src if s == "pv_xfused.hip" else os.path.join(b.CSRC, s)
    o = f"/tmp/diag_{s}.o"
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(s, []), "-I", b.CSRC, "-c", path, "-o", o])
    objs.append(o)
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
from photoverse_amd import _lib  # noqa: E402
_lib.LIB = lib
from photoverse_amd.ops import Recorder  # noqa: E402

dev = torch.device("cuda")
B, n, C, d, p = 16, 4096, 320, 40, 1
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
names = ["", "X loads issued", "DMA issue (K/V 0 + 3 stages)", "LayerNorm + drain", "(stamp gap)", "phase 1 (to_q)", "phase 2 (attention)",
         "exchange", "phase 3 (to_out) + epilogue"]
t = list(out)[:8]
for i in range(1, 8):
    print(f"{names[i]:32s} {t[i] - t[i - 1]:8d} ticks")
print(f"{'total (wave 0 of workgroup 300)':32s} {t[7] - t[0]:8d} ticks of s_memtime")
f = list(out)[8:13]
print("phase-1 stage 5:  reads H1 + MFMA H0 %d | vmcnt wait + barrier %d | issue stage %d | read H0 + MFMA H1 %d" % (f[1] - f[0], f[2] - f[1], f[3] - f[2], f[4] - f[3]))
